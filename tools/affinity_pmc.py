"""Target for rocprofv3 passes over the N x N affinity build (materialised fp32 wij) at a batch where its bytes are
meaningful:  python tools/affinity_pmc.py [B=1024] [reps=3] [N ...=200 400]
Prints the HIP-event time per launch as well (no profiler needed for that)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relativepose_amd import rpmodule, synth
from relativepose_amd.params import FINAL_PARAMS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Ns = [int(a) for a in sys.argv[3:]] or [200, 400]
dev = torch.device("cuda", 0)
para = rpmodule.opts(*FINAL_PARAMS[os.environ.get("RELPOSE_AFF_PARAMS", "suncg")][0])     # (scannet: sigmaFeat 0.0115 -> a dense wij window)
if os.environ.get("RELPOSE_AFF_SEL"):          # A/B: force a kernel variant (relpose_set_tuning) for the whole run
    from relativepose_amd import _lib
    _lib.lib().relpose_set_tuning(_lib.TUNE_KEYS["affinity_kernel"], _lib.AFFINITY_KERNELS[os.environ["RELPOSE_AFF_SEL"]])
for N in Ns:
    base = [synth.make_match_case(N, 5000 + b)[:2] for b in range(32)]
    kp = rpmodule.pack_keypoints([base[i % 32] for i in range(B)], dev)
    f_s, w_s, f_t, w_t, ns_, nt_ = kp[2], kp[3], kp[6], kp[7], kp[8], kp[9]
    for want in (True, False):
        outs = rpmodule.affinity_topk_buffers(B, N, N, para.topK, dev, want_wij=want)
        rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=want, out=outs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=want, out=outs)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        by = ((N + N) * 33 * 4 + (N * N * 4 if want else N * 5 * 12)) * B
        print(f"N={N} B={B} wij={'written' if want else 'fused'} {ms * 1e3:.1f} us/launch  {by / 1e6:.1f} MB algorithmic  {by / ms / 1e6:.1f} GB/s", flush=True)
