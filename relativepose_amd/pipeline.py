"""The per-pair recurrent loop of the reference's evaluation driver
(evaluation.py:203-284; library form RPModule/rpmodule.py:569-662), batched over
B scan pairs and kept on the device from the input panoramas to the 4x4 poses:

    for step in range(alterStep):                     # 3 recurrent levels
        warp the other view with the current pose     util.warping
        SCNet on [2B,16,h,4h]                          (BatchNorm groups = pairs)
        compose + sample keypoint primitives          evaluation.py:246-253, getMatchingPrimitive
        spectral matching + robust fit -> R_hat        RelativePoseEstimation_helper

Keypoints: either an input (keypoints="given": ONE fixed set per view for all levels -- the BASELINE benchmark workload) or
derived at EVERY level from that level's feature maps like the reference does (keypoints="reference": rputil.getKeypoint /
getKeypoint_kinect behind their SIFT detector -- SIFT detections are level-invariant and given once per view, the feature-guided
augmentation + random fill + weights run on the device per level, csrc/keypoints.hip; evaluation.py:278 -> rpmodule.py:511-533).
"""
import numpy as np

from . import _lib, rpmodule, util


class RelativePosePipeline:
    _chain_nets = False
    _net_stream = None
    _net_streams = None

    def __init__(self, net, dataset="suncg", mask_method="second", sigmas=None, alter_steps=3, completion=1, max_edges=0, compose=0, outputs="all",
                 self_stream_cache=True, tail_overlap=True, net_priority=None, loop_fit_cluster=1, keypoints="given"):
        self.net = net
        # "given": prepare(pts, ptw) fixes the keypoints of every view for all levels.  "reference": prepare(sift=...) takes the views' SIFT
        # detections (panorama coordinates, rputil.map_detections) and every level derives its keypoints from its own feature maps, as
        # rputil.getKeypoint does in the reference (the np.random draws pre-drawn per pair and level: prepare(kp_seeds=...)).
        if keypoints not in ("given", "reference"):
            raise ValueError("keypoints must be 'given' or 'reference'")
        self.keypoints = keypoints
        # the masked own views (channels 0:8 of the net input) are written once per pass and only the warped partner view changes from
        # level to level (evaluation.py:217-242): levels >= 1 reuse level 0's self-view encoder streams (SCNet.forward(self_tag=...),
        # bitwise the same output).  False: every level recomputes them, like the reference.
        self.self_stream_cache = self_stream_cache
        # "all": SCNet computes every output like the reference; "pose": only the heads this loop reads (normal, depth, features:
        # RELPOSE_FWD_POSE_OUTPUTS) -- the same poses bit for bit, the completed rgb / semantic maps are not produced (opt-in)
        self.outputs = outputs
        self.dataset = dataset
        self.mask_method = mask_method
        self.alter_steps = alter_steps
        self.completion = completion
        self.max_edges = max_edges
        self.compose = compose          # 0: evaluation.py:250-251 (the driver), 1: rpmodule.py:633-634 (the library loop)
        if sigmas is None:
            o = rpmodule.opts()
            sigmas = [[o.sigmaAngle1, o.sigmaAngle2, o.sigmaDist, o.sigmaFeat]] * alter_steps
        self.sigmas = np.asarray(sigmas, dtype=np.float64).reshape(-1, 4)
        self.feat_off = 7 + net.snumclass
        self._slot_streams = []
        # tail_overlap=False: the whole forward on the SCNet stream in the pipelined modes (A/B switch; the product reads no environment
        # variable).  net_priority: HIP stream priority of the SCNet stream, None = -1 (high) when the tail overlaps, else 0.
        self.tail_overlap = bool(tail_overlap)
        self.net_priority = net_priority
        # workgroups per scan pair in the fit while several batches are in flight (run_pipelined; 1 = no helper workgroups: they take CUs from the
        # other slot's convolutions; A/B switch).  It travels WITH every matcher call (RelposeMatchArgs::fit_cluster): two pipelines in
        # one process with different settings do not interfere (round 5 set a process-wide knob around the loop here).
        self.loop_fit_cluster = int(loop_fit_cluster)

    def prepare(self, rgb, norm, depth, pts, ptw, device, keep_host=False, sift=None, kp_seeds=None):
        """Host arrays (dataset dict layout: rgb/norm [B,2,3,h,4h], depth [B,2,h,4h] f32; pts [B,2,N,2],
        ptw [B,2,N] f64) -> device-resident state.  Not part of the timed region.
        keypoints="reference": pts / ptw are ignored (None); sift = [(source detections [n,2], target detections [m,2])] * B in panorama
        coordinates (rputil.map_detections), kp_seeds [B][alter_steps] = the np.random seed of pair b's getKeypoint call at each level
        (default: 7919 * b + level)."""
        import torch
        B, _, _, h, w = rgb.shape
        if self.keypoints == "reference":
            return self._prepare_reference(rgb, norm, depth, device, keep_host, sift, kp_seeds)
        pts = np.asarray(pts, dtype=np.float64)
        ptw = np.asarray(ptw, dtype=np.float64)
        N = pts.shape[2]
        npts = np.full((B, 2), N, dtype=np.int32)
        if not self.completion:                       # rpmodule.py:534-537: keep observed-region keypoints only
            p2, w2 = np.zeros_like(pts), np.zeros_like(ptw)
            for b in range(B):
                for v in range(2):
                    k = ptw[b, v] == 1
                    npts[b, v] = int(k.sum())
                    p2[b, v, :npts[b, v]] = pts[b, v][k]
                    w2[b, v, :npts[b, v]] = ptw[b, v][k]
            pts, ptw = p2, w2
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
        st = {"B": B, "h": h, "N": N}
        st["rgb"] = t(rgb.reshape(2 * B, 3, h, w), torch.float32)
        st["norm"] = t(norm.reshape(2 * B, 3, h, w), torch.float32)
        st["depth"] = t(depth.reshape(2 * B, h, w), torch.float32)
        st["pts"] = t(pts.reshape(2 * B, N, 2), torch.float64)
        st["npts"] = t(npts.reshape(2 * B), torch.int32)
        st["w_s"] = t(ptw[:, 0], torch.float64)
        st["w_t"] = t(ptw[:, 1], torch.float64)
        st["ns"] = t(npts[:, 0], torch.int32)
        st["nt"] = t(npts[:, 1], torch.int32)
        st["eye"] = torch.eye(4, dtype=torch.float64, device=device).repeat(B, 1, 1).contiguous()
        # the batch's network input and output live as long as the prepared batch: allocated HERE, not at first use inside a serving loop
        # (1.7 + 5.7 GB per batch at 320x1280: a hipMalloc of that size in the loop stalls every stream -- bench configs[4] ran at 613, 851
        # or 965-983 pairs/s depending on which of the rotated batches the caching allocator could serve from freed blocks)
        st["x"] = torch.empty(2 * B, 16, h, w, dtype=torch.float32, device=device)
        st["f"] = torch.empty(2 * B, self.net.out_channels, h, w, dtype=torch.float32, device=device)
        if keep_host:      # pinned host copies of the per-batch inputs, for upload_inputs (PCIe-inclusive timing)
            st["host"] = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dt).pin_memory() for k, a, dt in (
                ("rgb", rgb.reshape(2 * B, 3, h, w), torch.float32), ("norm", norm.reshape(2 * B, 3, h, w), torch.float32),
                ("depth", depth.reshape(2 * B, h, w), torch.float32), ("pts", pts.reshape(2 * B, N, 2), torch.float64))}
        return st

    def _prepare_reference(self, rgb, norm, depth, device, keep_host, sift, kp_seeds):
        """prepare() for keypoints="reference": the feature-independent half of every level's getKeypoint call is drawn here, on the host, in the
        reference's np.random call order (rputil.keypoint_plan) and uploaded as query points + slot tables; one keypoint capacity L for all levels."""
        import torch
        from . import rputil
        B, _, _, h, w = rgb.shape
        if sift is None or len(sift) != B:
            raise ValueError("keypoints='reference': prepare(sift=[(source detections, target detections)] * B) is required")
        kind = self.mask_method
        tabs = []
        for lvl in range(self.alter_steps):
            plans = []
            for b in range(B):
                seed = int(kp_seeds[b][lvl]) if kp_seeds is not None else 7919 * b + lvl
                plans.append(rputil.keypoint_plan(sift[b][0], sift[b][1], kind, h, w, np.random.RandomState(seed)))
            tabs.append(rputil.keypoint_tables(plans, h, w))
        L = max(t["L"] for t in tabs)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
        st = {"B": B, "h": h, "N": L}
        st["rgb"] = t(rgb.reshape(2 * B, 3, h, w), torch.float32)
        st["norm"] = t(norm.reshape(2 * B, 3, h, w), torch.float32)
        st["depth"] = t(depth.reshape(2 * B, h, w), torch.float32)
        st["kp"] = [rputil.upload_keypoint_tables(tb, device, L) for tb in tabs]
        nb = max(_lib.lib().relpose_keypoints_reference_workspace_bytes(max(tb["nq"], 1), h, w, tb["topk"]) for tb in tabs)
        st["kp_ws"] = torch.empty(nb, dtype=torch.uint8, device=device)
        st["eye"] = torch.eye(4, dtype=torch.float64, device=device).repeat(B, 1, 1).contiguous()
        st["x"] = torch.empty(2 * B, 16, h, w, dtype=torch.float32, device=device)
        st["f"] = torch.empty(2 * B, self.net.out_channels, h, w, dtype=torch.float32, device=device)
        if keep_host:
            st["host"] = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dt).pin_memory() for k, a, dt in (
                ("rgb", rgb.reshape(2 * B, 3, h, w), torch.float32), ("norm", norm.reshape(2 * B, 3, h, w), torch.float32),
                ("depth", depth.reshape(2 * B, h, w), torch.float32))}
        return st

    def _level_keypoints(self, st, f, step):
        """(pts [2B,N,2], npts [2B], w_s, w_t [B,N], ns, nt [B]) of this level: the prepared set, or -- keypoints="reference" -- derived from this
        level's feature maps on the device (rputil.getKeypoint behind the detector, relpose_keypoints_reference)."""
        if self.keypoints != "reference":
            return st["pts"], st["npts"], st["w_s"], st["w_t"], st["ns"], st["nt"]
        from . import rputil
        B, N = st["B"], st["N"]
        # completion = 0 (the 'ours_nc' method, evaluation.py:74): only the observed-region keypoints take part (rpmodule.py:534-537)
        pts, w, npts = rputil.keypoints_reference_dev(f, self.feat_off, st["kp"][min(step, len(st["kp"]) - 1)], self.mask_method, L=N, workspace=st["kp_ws"],
                                                      observed_only=not self.completion)
        w = w.view(B, 2, N)
        n2 = npts.view(B, 2)
        return pts, npts, w[:, 0].contiguous(), w[:, 1].contiguous(), n2[:, 0].contiguous(), n2[:, 1].contiguous()

    @staticmethod
    def upload_inputs(st, copy_stream):
        """Re-upload the batch's panoramas + keypoints from pinned host memory on `copy_stream` (async), ordered after
        the previous use of the device buffers and before the batch's next use (both on st["stream"])."""
        import torch
        ms = st["stream"] if "stream" in st else torch.cuda.current_stream()
        if copy_stream is None:
            # on the batch's own (slot) stream: it is idle between the previous batch's last matcher and this batch's first
            # warp, the other slot's forward runs meanwhile, and no fifth stream has to share a hardware queue (run_pipelined)
            with torch.cuda.stream(ms):
                for k, src in st["host"].items():
                    st[k].copy_(src, non_blocking=True)
            return
        copy_stream.wait_stream(ms)
        with torch.cuda.stream(copy_stream):
            for k, src in st["host"].items():
                st[k].copy_(src, non_blocking=True)
        ms.wait_stream(copy_stream)

    def run_interleaved(self, states):
        """Several prepared batches software-pipelined on their own HIP streams: the SCNet forwards are chained
        by events (one at a time, each owning the whole GPU: A0 B0 A1 B1 ...), so the launch-bound matcher phase
        of one batch (hundreds of few-microsecond kernels) always runs UNDER the MFMA-bound SCNet phase of the
        next one instead of next to another matcher.  Per-pair results are identical to `run` (same kernels,
        same data).  Returns [(pose, status)]."""
        import torch
        cur = torch.cuda.current_stream()
        self._chain_nets = len(states) > 1
        self._ensure_net_stream(states)
        for st in states:
            if "stream" not in st:
                st["stream"] = torch.cuda.Stream()
            st["stream"].wait_stream(cur)
        gens = [self._run_gen(st) for st in states]
        out = [None] * len(states)
        live = list(range(len(states)))
        while live:
            for i in list(live):
                with torch.cuda.stream(states[i]["stream"]):
                    try:
                        next(gens[i])
                    except StopIteration as e:
                        out[i] = e.value
                        live.remove(i)
        for st in states:
            cur.wait_stream(st["stream"])
        cur.wait_stream(self._net_stream)
        self._chain_nets = False
        return [(o[0], o[1]) for o in out]

    def _ensure_net_stream(self, states=None):
        # created BEFORE the per-batch streams: on this runtime the tiny kernels of a later-created stream are
        # dispatched promptly next to an earlier-created stream's big grids, but not the other way round
        # (measured: matcher phase 16 ms vs 29 ms under a concurrent forward, profiles/r01_overlap.txt).  Round-2 experiments on the
        # scheduling of this stream: high stream priority while the matcher still took 10 ms (it starved and became critical: 394 ->
        # 365-377 pairs/s; see below for today's rule) and two SCNet streams whose forwards may overlap (394 -> 358-360: the two
        # working sets fight over L2)
        import torch
        # HIP stream priority of the SCNet stream (the `net_priority` argument overrides; -1 = high: conv workgroups are dispatched ahead of
        # the slot streams' kernels, which then fill the holes -- the drain of every conv launch, barrier stalls).  With the forwards'
        # head / tail on the slot streams this is worth +4 % at 200 keypoints (499 -> 518 pairs/s, round 3).  Round 2 limited it to
        # <= 256 keypoints per view (at 400 the deprioritised slot-stream chain tail -> matcher -> warp -> head became critical: -3 %);
        # with round 3's matcher and level-0 plan that is gone (configs[2]: 482 vs 484 pairs/s), so it is on whenever the tail overlaps.
        if self._net_streams is None:
            self._net_streams = {0: torch.cuda.Stream(priority=0), -1: torch.cuda.Stream(priority=-1)}
        prio = self.net_priority if self.net_priority is not None else (-1 if self.tail_overlap else 0)
        new_stream = self._net_streams[-1 if prio < 0 else 0]
        if self._net_stream is not None and new_stream is not self._net_stream:
            new_stream.wait_stream(self._net_stream)            # forwards of the previous call stay ordered before this call's
        self._net_stream = new_stream
        self._net_stream.wait_stream(torch.cuda.current_stream())

    def run_pipelined(self, states, steps, on_result=None, depth=None, before_batch=None, provider=None):
        """`steps` consecutive batches through the hot path with `depth` of them in flight (a serving loop;
        default depth = len(states)): batch k uses the prepared buffers and the HIP stream of
        states[k % len(states)], so with more prepared states than batches in flight the loop rotates through
        distinct inputs.  SCNet forwards are chained by events exactly as in `run_interleaved`, so while batch k
        is in its launch-bound matcher phase the GPU runs the SCNet forward of batch k+1 -- both at the FULL
        batch size (splitting one batch over streams halves the SCNet batch and costs ~13 % conv efficiency).
        before_batch(k, state) is called (under the state's stream) right before batch k is started -- the
        bench uploads the batch's inputs there.  on_result(k, pose, status) -> value is called on the caller's
        stream as soon as batch k is complete.  Returns the per-batch results in order."""
        import torch
        cur = torch.cuda.current_stream()
        if provider is not None:
            states = []
            nst = max(1, steps)
            depth = max(1, min(2 if depth is None else depth, nst))
        else:
            nst = len(states)
            depth = nst if depth is None else max(1, min(depth, nst))
        self._chain_nets = depth > 1
        self._ensure_net_stream(states)
        # One HIP stream per IN-FLIGHT SLOT, not per prepared batch: the runtime multiplexes streams onto 4 hardware queues, and with
        # net + default + 4 batch streams two batch streams share a queue -- the input preparation of a new batch then sits behind
        # the other batch's last matcher (which waits for ITS forward) in that queue, and the SCNet stream idles for a matcher
        # (4.5 ms per 2 batches, profiles/r02_overlap.txt).  net + default + 2 slots = 4 streams = 4 queues; a new batch is
        # ordered behind the finished batch of its own slot, whose matcher ran under the other slot's forward long before.
        while len(self._slot_streams) < depth:
            self._slot_streams.append(torch.cuda.Stream())
        for ss in self._slot_streams[:depth]:
            ss.wait_stream(cur)
        for st in states:
            if "stream" in st:
                st["stream"].wait_stream(cur)
        results = [None] * steps
        live, nxt = {}, 0
        # batches in flight = a throughput loop: the fit's helper workgroups (a latency tool for a lone small batch)
        # would take CUs from the other slot's convolutions, so the matcher calls enqueued here (_run_gen) carry fit_cluster = loop_fit_cluster
        while nxt < steps or live:
            for slot in range(depth):
                if slot not in live and nxt < steps:
                    ss = self._slot_streams[slot]
                    if provider is not None:
                        st = provider(nxt)               # (allocates + uploads on the caller's stream)
                        ss.wait_stream(cur)
                    else:
                        st = states[nxt % nst]
                    if st.get("stream") is not None and st["stream"] is not ss:
                        # the buffers' previous use (another slot / run_interleaved): the batch that used them recorded its completion -- wait for
                        # THAT, not for whatever else has been queued on its slot stream since (with 4 rotating batches and 3 in flight a state
                        # changes slot every time: waiting for the whole stream serialised the slots, 592 vs 634 pairs/s at configs[2])
                        if st.get("done_ev") is not None:
                            ss.wait_event(st["done_ev"])
                        else:
                            ss.wait_stream(st["stream"])
                    st["stream"] = ss
                    if before_batch is not None:
                        with torch.cuda.stream(st["stream"]):
                            before_batch(nxt, st)
                    live[slot] = (nxt, st, self._run_gen(st))
                    nxt += 1
                if slot not in live:
                    continue
                k, st, gen = live[slot]
                done = None
                with torch.cuda.stream(st["stream"]):
                    try:
                        next(gen)
                    except StopIteration as e:
                        done = e.value
                if done is not None:
                    del live[slot]
                    # the batch's buffers are free again once this point of its stream is reached (a copy stream that refills them for a
                    # later batch waits for THIS event, not for whatever else is queued on the slot stream: bench.py's upload look-ahead)
                    st["done_ev"] = torch.cuda.Event()
                    st["done_ev"].record(st["stream"])
                    pose, status = done[0], done[1]
                    if on_result is not None:
                        cur.wait_stream(st["stream"])
                        pose.record_stream(cur); status.record_stream(cur)
                        results[k] = on_result(k, pose, status)
                    else:
                        if provider is not None:
                            # a provider's state was allocated on the caller's stream and is dropped right here: order the caller's stream behind
                            # the batch, or the caching allocator may hand the blocks to the next provider() upload while kernels still read them
                            cur.wait_stream(st["stream"])
                            cur.wait_stream(self._net_stream)
                        results[k] = (pose, status)
        for st in states:
            if "stream" in st:
                cur.wait_stream(st["stream"])
        for ss in self._slot_streams[:depth]:
            cur.wait_stream(ss)
        cur.wait_stream(self._net_stream)
        self._chain_nets = False
        return results

    def _net_input(self, st):
        """The network input [2B,16,h,4h] of a prepared batch (kept across calls): channels 0:8 = the masked own
        view of every image (util.apply_mask layout), channels 8:16 are rewritten by every level's warp."""
        import torch
        B, h = st["B"], st["h"]
        view = util.build_view_dev(st["rgb"], st["norm"], st["depth"], self.mask_method)      # [2B,8,h,4h]
        x = st.get("x")
        if x is None:
            x = st["x"] = torch.empty(2 * B, 16, h, 4 * h, dtype=torch.float32, device=view.device)
        x[:, :8].copy_(view)
        # a fresh name for this content of channels 0:8 (whoever finds the same tag on its workspace may reuse the self-view streams)
        st["self_tag"] = self.net.new_self_tag() if self.self_stream_cache else 0
        # the network output of this batch, reused by every level and every pass (a torch.empty per level is 1.4 GB at 160x640 and 5.7 GB at
        # 320x1280: blocks that the caching allocator cannot always recycle in time across streams -> hipMalloc inside the loop)
        f = st.get("f")
        if f is None or f.shape[1] != self.net.out_channels:
            st["f"] = torch.empty(2 * B, self.net.out_channels, h, 4 * h, dtype=torch.float32, device=view.device)
        return x

    def _run_gen(self, st):
        """`run` as a generator for the software pipelines: yields once per level, right after enqueueing the
        SCNet forward (so a batch's matcher + next warp are enqueued back to back, before the other batches')."""
        import torch
        B, h, N = st["B"], st["h"], st["N"]
        x = self._net_input(st)
        R_hat, status = st["eye"], None
        for step in range(self.alter_steps):
            # (level 0: the estimate IS the identity, and so is its inverse -- taken literally, not through the inversion kernel, because
            # the zero-warp plan below relies on BOTH warps of a pair being the identity's all-zero view, util.py:95-96)
            inv = R_hat if step == 0 else util.pose_inverse_dev(R_hat)
            poses = torch.stack((inv, R_hat), 1).reshape(2 * B, 4, 4).contiguous()
            util.warp_pairs_dev(x, poses, self.dataset)       # x[:, 8:] = partner view warped by the pose estimate
            if self._chain_nets:
                # every SCNet forward of every batch in flight goes to ONE dedicated stream, in enqueue order
                ms, ns = torch.cuda.current_stream(), self._net_stream
                ns.wait_stream(ms)
                if self.tail_overlap:
                    # the convolutions on the SCNet stream, the HBM-bound tail (heads + resize, 1.9 ms) on this batch's own stream: the
                    # SCNet stream goes straight on to the other batch's forward, whose MFMA-bound convs overlap this tail.  Overlapping
                    # forwards need separate workspaces: one per stream (= per in-flight slot).
                    f = st["f"]
                    with torch.cuda.stream(ns):
                        self.net.forward(x, out=f, tail_stream=ms, ws_key=ms.cuda_stream, zero_warp=(step == 0), outputs=self.outputs,
                                         self_tag=st["self_tag"])
                else:
                    with torch.cuda.stream(ns):
                        f = self.net.forward(x, out=st["f"], zero_warp=(step == 0), outputs=self.outputs, self_tag=st["self_tag"])
                        done = torch.cuda.Event()
                        done.record()
                    ms.wait_event(done)
                yield                                            # one yield per level: the other batches enqueue theirs
            else:
                f = self.net.forward(x, out=st["f"], zero_warp=(step == 0), outputs=self.outputs, self_tag=st["self_tag"])
            kpts, knpts, w_s, w_t, ns, nt = self._level_keypoints(st, f, step)
            pc, nn, ft = util.sample_primitives_dev(f, self.feat_off, st["norm"], st["depth"], kpts, knpts,
                                                    self.mask_method, self.dataset, self.compose)
            pc, nn, ft = pc.view(B, 2, N, 3), nn.view(B, 2, N, 3), ft.view(B, 2, N, 32)
            para = rpmodule.opts(*self.sigmas[min(step, len(self.sigmas) - 1)])
            res = rpmodule.match_pairs(pc[:, 0].contiguous(), nn[:, 0].contiguous(), ft[:, 0].contiguous(), w_s,
                                       pc[:, 1].contiguous(), nn[:, 1].contiguous(), ft[:, 1].contiguous(), w_t,
                                       ns, nt, para, max_edges=self.max_edges, fit_cluster=self.loop_fit_cluster if self._chain_nets else 0)
            R_hat, status = res.pose, res.status
        return R_hat, status, None

    def run(self, st, R_forced=None, keep=None, primitives=None):
        """One pass of the hot path over the prepared batch.  Returns (pose [B,4,4] f64, status [B] i32,
        [pose after each step]).  R_forced: optional list of [B,4,4] tensors (teacher forcing, tests).
        primitives: a dict that receives the LAST level's matching primitives (pc, nn [B,2,N,3] f64, ft [B,2,N,32] f32: what the reference's
        tuning script caches per pair, trainRelativePoseModuleRecFD.py:207-208; tune.cache_primitives)."""
        import torch
        B, h, N = st["B"], st["h"], st["N"]
        x = self._net_input(st)                                                               # [2B,16,h,4h]
        R_hat = st["eye"]
        trace = []
        status = None
        for step in range(self.alter_steps):
            if R_forced is not None:
                R_hat = R_forced[step]
            # image 2b (source) gets target warped by inv(R), image 2b+1 (target) gets source warped by R
            inv = R_hat if (step == 0 and R_forced is None) else util.pose_inverse_dev(R_hat)
            poses = torch.stack((inv, R_hat), 1).reshape(2 * B, 4, 4).contiguous()
            util.warp_pairs_dev(x, poses, self.dataset)       # x[:, 8:] = partner view warped by the pose estimate
            # level 0 starts from the identity: util.warping returns zeros (util.py:95-96) for every image, which SCNet can exploit
            # (the batch's preallocated output buffer, unless the caller keeps every level's output: `keep` entries must not alias)
            f = self.net.forward(x, out=st["f"] if keep is None else None, zero_warp=(step == 0 and R_forced is None), outputs=self.outputs,
                                 self_tag=st["self_tag"])
            kpts, knpts, w_s, w_t, ns, nt = self._level_keypoints(st, f, step)
            pc, nn, ft = util.sample_primitives_dev(f, self.feat_off, st["norm"], st["depth"], kpts, knpts,
                                                    self.mask_method, self.dataset, self.compose)
            pc, nn, ft = pc.view(B, 2, N, 3), nn.view(B, 2, N, 3), ft.view(B, 2, N, 32)
            para = rpmodule.opts(*self.sigmas[min(step, len(self.sigmas) - 1)])
            res = rpmodule.match_pairs(pc[:, 0].contiguous(), nn[:, 0].contiguous(), ft[:, 0].contiguous(), w_s,
                                       pc[:, 1].contiguous(), nn[:, 1].contiguous(), ft[:, 1].contiguous(), w_t,
                                       ns, nt, para, max_edges=self.max_edges)
            R_hat, status = res.pose, res.status
            trace.append(R_hat)
            if primitives is not None:
                primitives.update(pc=pc, nn=nn, ft=ft, step=step, w_s=w_s, w_t=w_t, ns=ns, nt=nt)
            if keep is not None:
                keep.append({"x": x.clone(), "f": f, "pc": pc, "nn": nn, "ft": ft, "pts": kpts.view(B, 2, N, 2), "w_s": w_s, "w_t": w_t, "ns": ns, "nt": nt})
        return R_hat, status, trace
