"""CPU (needs only hipcc, which cross-compiles gfx950 without a GPU): properties of the GENERATED device code that the GPU tests can
only show as crashes or slow kernels.

* No irreducible control flow.  Round 3's helper-workgroup fit first had two thread-0 blocks in its claim loop; hipcc threaded thread 0
  from the end of the body into the next claim, the loop became irreducible (LLVM's `irr.guard` blocks), and wave 0's other lanes went
  through the barrier -- and read the claimed chunk index -- before lane 0 had written it: a memory fault on the GPU.  Barriers and
  irreducible regions do not mix, so no kernel of the library may contain one.
* The kernels the headline runs on keep their register budget: no VGPR spills in the fp32 tile kernels, conv1, the heads and the
  512-thread fit; the 1024-thread fit and the affinity tile kernel stay at 128 VGPRs (4 waves per SIMD)."""
import os
import re
import shutil
import subprocess

import pytest

from relativepose_amd import build as B

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("devasm")
    out = {}
    for src, extra in B.SOURCES:
        s = d / (src[:-4] + ".s")
        # -fno-discard-value-names: release clang drops IR value names, and FixIrreducible's blocks would print as %bb.N instead of irr.guard
        subprocess.check_call([HIPCC, f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", *extra, "--cuda-device-only", "-fno-discard-value-names", "-S",
                               "-o", str(s), os.path.join(B.CSRC, src)], stderr=subprocess.DEVNULL)
        out[src] = s.read_text()
    shutil.rmtree(d, ignore_errors=True)
    return out


def _kernels(txt):
    """kernel name -> {vgpr_count, vgpr_spill_count, ...} from the amdhsa metadata of an assembly listing"""
    res = {}
    # (the kernel-level .name is directly followed by .private_segment_fixed_size; with value names kept, arguments have .name entries too)
    for m in re.finditer(r"\.name:\s+(\S+)\n(\s+\.private_segment_fixed_size:.*?)\.wavefront_size", txt, re.S):
        res[m.group(1)] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", m.group(2))}
    return res


IRREDUCIBLE_KERNEL = r"""
#include <hip/hip_runtime.h>
// two entries into the cycle {L1, L2}: an irreducible loop (positive control of the check below)
extern "C" __global__ void irreducible_control(int* p, int n) {
    int i = threadIdx.x;
    if (p[i & 3] > 0) goto L2;
L1:
    i += p[i & 7];
    __syncthreads();
L2:
    i += 3;
    p[i & 15] = i;
    if (i < n) goto L1;
}
"""


def test_irreducible_control_flow_check_can_fail(tmp_path):
    """Positive control: a kernel with a known irreducible loop must trip the marker the next test greps for (without
    -fno-discard-value-names the marker never appears and that test could not fail: ADVICE r3)."""
    src = tmp_path / "irr.hip"
    src.write_text(IRREDUCIBLE_KERNEL)
    out = tmp_path / "irr.s"
    subprocess.check_call([HIPCC, f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "--cuda-device-only", "-fno-discard-value-names", "-S",
                           "-o", str(out), str(src)], stderr=subprocess.DEVNULL)
    assert "irr.guard" in out.read_text()


def test_no_kernel_has_irreducible_control_flow(device_asm):
    for src, txt in device_asm.items():
        assert "s_barrier" in txt or src == "geometry.hip"
        assert "irr.guard" not in txt, f"{src}: hipcc produced an irreducible region (FixIrreducible's irr.guard blocks): restructure the loop"


def test_headline_kernels_keep_their_register_budget(device_asm):
    k = {}
    for txt in device_asm.values():
        k.update(_kernels(txt))
    assert len(k) > 100

    def find(*parts):
        hits = [n for n in k if all(p in n for p in parts)]
        assert hits, parts
        return hits

    spill_free = [("conv_s2_tile_kernelILi2ELi2ELi16ELb0ELi0E",), ("conv_s2_tile_kernelILi1ELi4ELi8ELb1ELi0E",), ("conv_s2_strip_kernelILi4ELi0E",),
                  ("deconv_tile_kernelILi1ELi1ELi4ELi16ELb0ELi0E",), ("deconv_tile_kernelILi1ELi2ELi4ELi16ELb0ELi0E",),
                  ("deconv_tile_kernelILi1ELi1ELi4ELi8ELb1ELi0E",), ("conv1_mfma_kernel",), ("heads_kernelILi15E",), ("heads_kernelILi21E",),
                  ("resize_out_kernel",), ("conv_igemm_kernelILi2ELi2ELi2ELi2ELb1ELb0ELi0E",),
                  ("conv_igemm_kernelILi2ELi2ELi2ELi2ELb1ELb1ELi0E",)]
    for parts in spill_free:
        for n in find(*parts):
            assert k[n]["vgpr_spill_count"] == 0, (n, k[n])
    for n in find("affinity_tile_kernel"):
        assert k[n]["vgpr_count"] <= 128, (n, k[n])
    # the fit's 512-thread kernel: at most a handful of long-lived values parked in scratch AROUND the eigen-solves (round 6: 17 dwords -- the
    # IRLS's cached correspondence, stored before the first Lanczos cycle and reloaded behind the last; none inside an edge pass or the
    # re-orthogonalisation: profiles/r06_matcher.txt measures the kernel 21 % faster than the spill-free round-5 one)
    for n in find("fit_pair_kernelILi512ELi0E"):
        assert k[n]["vgpr_spill_count"] <= 24, (n, k[n])
    # the fit's large kernel (more than 1024 correspondences: 768 threads since round 6) keeps three waves per SIMD
    for n in find("fit_pair_kernelILi768E"):
        assert k[n]["vgpr_count"] <= 170, (n, k[n])
    # the affinity tile kernel keeps no array in scratch memory (round 4: a select between elements of the source-row array had turned into
    # a dynamically indexed load and sent the whole array -- 128+ bytes per lane -- to scratch, re-read in every pooled evaluation); the
    # fused variant does not spill at all, the materialising one at most a few registers around its (cold) dense-window path
    for n in find("affinity_tile_kernel"):
        assert k[n]["vgpr_spill_count"] <= 8 and k[n]["private_segment_fixed_size"] <= 40, (n, k[n])
    for n in find("affinity_tile_kernelILb0E"):
        assert k[n]["vgpr_spill_count"] == 0 and k[n]["private_segment_fixed_size"] == 0, (n, k[n])
    # three workgroups of the fp32 tile kernels per CU: <= 168 VGPRs
    for parts in spill_free[:6]:
        for n in find(*parts):
            assert k[n]["vgpr_count"] <= 168 or "Li1ELi2ELi4ELi16" in n, (n, k[n])
