#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE passes) into
profiles/pmc_traffic.json: HBM-side bytes per launch of the instrumented kernels.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: pmc_summary.py <dir with *_counter_collection.csv files ...>"""
import csv
import glob
import json
import os
import sys

NAMES = {
    "cheb_a00_step_kernel": "cheb_step_A00",
    "spmv_full_kernel": "spmv_full_block3x3",
    "spmv3_full_kernel": "spmv3_full_block4x4",
    "asm_kernel<1": "asm_residual_jacobian",
    "asm_kernel<2": "asm_residual_only",
    "asm3q_kernel<1": "asm3_residual_jacobian",
    "moments_kernel": "tau_moments",
    "moments3_kernel": "tau_moments_tet",
    "spmv_a01_resid_kernel": "coupling_product_A01",
    "fused_up_sell_kernel<HIP_vector_type<double, 2": "amg_up0_velocity_2rhs",
    "fused_up_sell_kernel<double": "amg_up0_pressure",
    "fused_down_kernel<16, float, HIP_vector_type<double, 2": "amg_down0_velocity_2rhs",
    "fused_down_kernel<16, float, double": "amg_down0_pressure",
    "sell_cheb2_scale_kernel": "cheb2_scale_H",
    "multidot_kernel": "multidot",
    "gs_update_normalize_kernel": "gs_update_normalize",
}


def main(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row.get("Kernel_Name", "")
                key = next((v for k, v in NAMES.items() if k in kn), None)
                if key is None:
                    continue
                cname, val = row.get("Counter_Name"), float(row.get("Counter_Value", 0))
                # only full-size dispatches of the multi-level kernels (level 0): largest grid
                acc.setdefault((key, cname), []).append((int(row.get("Grid_Size", 0)), val))
    out = {}
    for (key, cname), vals in acc.items():
        gmax = max(g for g, _ in vals)
        v = [x for g, x in vals if g == gmax]
        mean = sum(v) / len(v)
        e = out.setdefault(key, {"fetch_bytes": None, "write_bytes": None, "launches_sampled": len(v)})
        if cname == "FETCH_SIZE":
            e["fetch_bytes"] = 2.0 * mean * 1024.0
        elif cname == "WRITE_SIZE":
            e["write_bytes"] = mean * 1024.0
    for e in out.values():
        if e["fetch_bytes"] is not None and e["write_bytes"] is not None:
            e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flat = {k: v.get("hbm_bytes") for k, v in out.items()}
    sys.path.insert(0, root)
    import bench  # the fingerprint of the kernel sources these counters were taken with: bench.py refuses a stale summary
    json.dump({"kernel_source_sha16": bench.kernel_source_sha16(), "per_launch_bytes": flat, "detail": out,
               "note": "FETCH_SIZE x2 (gfx950 correction), KiB->bytes; level-0 dispatches only"},
              open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(flat, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
