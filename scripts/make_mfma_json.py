"""Build profiles/rNN_pmc_mfma.json from the MFMA counter passes of profile_round.sh.

    python3 scripts/make_mfma_json.py <prof dir> <out json>

Per kernel (averages over its dispatches): duration, SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipes were
busy, summed over the 1024 SIMDs), GRBM_GUI_ACTIVE (active cycles, summed over the 8 XCDs: the value is 8 x the
kernel's duration in shader cycles, which also gives the clock the kernel actually ran at) and
mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs), the fraction of SIMD-cycles with a busy matrix pipe (ROCm 7.2
ships no gfx950 derived counters, MI355X_MICROARCH.md); plus the SQ wave-cycle split (wait / issue-stall / active)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic_json import first_db, per_kernel   # noqa: E402

SIMDS = 256 * 4
XCDS = 8


def short(name):
    # (the f32 direct kernels appear in every trace since round 5: vpk_cnn_load's calibration forwards run them once at batch 1;
    #  their per-launch averages there say nothing about a batch of 102 and are left out)
    for tag, s in (("conv_pieces_kernelILi5E", "conv_pieces_kernel<5> (conv2)"),
                   ("conv_pieces_kernelILi3ELi4ELi2ELi2E", "conv_pieces_kernel<3, 64 x 8 tiles> (conv4)"),
                   ("conv_pieces_kernelILi3E", "conv_pieces_kernel<3> (conv3, conv5)"),
                   ("conv1_pieces_kernel", "conv1_pieces_kernel (conv1+norm1+pool1)"),
                   ("dense_pieces_kernel", "dense_pieces_kernel (fc6, fc7: average of the two launches)"),
                   ("conv5x5_winograd_kernel", "conv5x5_winograd_kernel (conv2)"),
                   ("conv3x3_winograd_kernel", "conv3x3_winograd_kernel (conv3/4/5)"),
                   ("lrn5_pool3s2_planes", "lrn5_pool3s2_planes_kernel (norm2+pool2)"),
                   ("em_batch_kernel", "em_batch_kernel")):
        if tag in name:
            return s
    return None


def main(prof, out):
    res = {"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                   "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE (own pass); mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs * 1024 SIMDs); shader_clock = GUI_ACTIVE / 8 / duration"}
    for tag, label in (("cnn_mfma", "cnn_alone_B102"), ("yud_mfma", "bench_yud_102")):
        try:
            k = per_kernel(first_db(os.path.join(prof, tag)))
        except SystemExit:
            continue
        sec = {}
        for name, c in k.items():
            s = short(name)
            if s is None:
                continue
            busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            sec[s] = {"avg_ms": c["_ms"], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
                      "mfma_util": busy / (gui / XCDS * SIMDS) if gui else None,
                      "shader_clock_ghz": gui / XCDS / (c["_ms"] * 1e6) if gui else None,
                      "wave_cycles_wait_frac": c.get("SQ_WAIT_ANY", 0.0) / wc,
                      "wave_cycles_issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                      "wave_cycles_active_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        res[label] = sec
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
