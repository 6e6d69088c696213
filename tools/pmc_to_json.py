#!/usr/bin/env python3
"""profiles/layer_kernel_pmc.json from the per-kernel means a PMC run left (tools/profile_bench.sh -> pmc_summary.txt):
python3 tools/pmc_to_json.py gpurun_out/<tag>/pmc_summary.txt profiles/<archived summary name> > profiles/layer_kernel_pmc.json
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md)."""
import json
import sys

VARIANTS = {   # template arguments <NU, KX, L0, GRU, SAVE, HEAD> of the bench shape's three launches per reverse step
    "plain": "td_rnn_layer_kernel<8, 32, false, false, false, false>",
    "head": "td_rnn_layer_kernel<8, 32, false, false, false, true>",
    "l0": "td_rnn_layer_kernel<8, 1, true, false, false, false>",
}
rows, cur = {}, None
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        cur = line.strip()
        rows[cur] = {}
    elif cur:
        name, val = line.split()[:2]
        rows[cur][name] = float(val)


def entry(kernel):
    r = next(v for k, v in rows.items() if kernel in k)
    hbm = int((2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024)
    return {"kernel": kernel, "fetch_size_kb": r["FETCH_SIZE"], "write_size_kb": r["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
            "mfma_busy_cycles": r["SQ_VALU_MFMA_BUSY_CYCLES"], "grbm_gui_active_sum_over_xcds": r["GRBM_GUI_ACTIVE"],
            "mfma_util": r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] * 128),
            "valu_mfma_coexec_share_of_mfma_busy": r["SQ_VALU_MFMA_COEXEC_CYCLES"] / r["SQ_VALU_MFMA_BUSY_CYCLES"],
            "wait_any_share_of_wave_cycles": r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"]}


out = {"batch": 16384, "seq_len": 272, "hidden": 256, "variants": {k: entry(v) for k, v in VARIANTS.items()},
       "note": "rocprofv3 --pmc passes of tools/pmc_passes.sh (SQ_* / FETCH_SIZE GRBM_GUI_ACTIVE / WRITE_SIZE, one run each) over "
               "bench.py --timesteps 4 --probe-steps 2; means per launch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE "
               "summed over the 8 XCDs x 128 SIMDs per XCD); source " + (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])}
out.update(out["variants"]["plain"])      # the launch bench.py's roofline probe times (tensor-level forward: separate head)
json.dump(out, sys.stdout, indent=1)
print()
