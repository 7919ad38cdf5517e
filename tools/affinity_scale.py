"""affinity_topk_kernel<true> (materialised fp32 wij) bandwidth vs batch size: algorithmic bytes / event time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relativepose_amd import synth, rpmodule
from relativepose_amd.params import FINAL_PARAMS
SUNCG_SIGMAS = FINAL_PARAMS["suncg"]
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
base = [synth.make_match_case(N, 5000 + b)[:2] for b in range(32)]
para = rpmodule.opts(*SUNCG_SIGMAS[0])
for B in (32, 256, 1024, 4096):
    cases = [base[i % 32] for i in range(B)]
    kp = rpmodule.pack_keypoints(cases, dev)
    f_s, w_s, f_t, w_t, ns_, nt_ = kp[2], kp[3], kp[6], kp[7], kp[8], kp[9]
    for want in (True, False):
        for _ in range(3):
            rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=want)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        reps = 10
        for _ in range(reps):
            rpmodule.affinity_topk(f_s, w_s, f_t, w_t, ns_, nt_, para, want_wij=want)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        by = ((N + N) * 33 * 4 + (N * N * 4 if want else N * 5 * 12)) * B
        print(f"N={N} B={B:5d} wij={'written' if want else 'fused  '} {ms*1e3:9.1f} us  {by/1e6:9.1f} MB  {by/ms/1e6:8.1f} GB/s  ({by/ms/1e6/8000*100:.1f}% of 8 TB/s)")
