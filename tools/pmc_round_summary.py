"""Turn the PMC passes of tools/profile_round.sh into the small JSON files bench.py reads for `roofline.traffic`
(gpurun_out/<tag>_pmc_*.json -> copy into profiles/).  traffic = 2 * FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE counts a
128-byte fabric read as 64 bytes for 16-byte-per-lane streams (MI355X_MICROARCH.md, HBM section); both counters are in KiB."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
OUT = "gpurun_out"


def per_kernel(pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


for prec, sfx, head_kernel, conv_kernel, esz in (("fp32", "", "head_bwd_pc_kernel", ("conv3x3_fwd_s3_kernel", "conv3x3_mfma_kernel"), 4),
                                                 ("bf16", "_bf16", "head_bwd_bf16_coop4_kernel", "conv3x3_cl_kernel", 2)):
    step = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for n, cs in per_kernel(f"{OUT}/pmc_{prec}_{c}/**/*counter_collection.csv").items():
            if c in cs:
                step[n][c] = (mean(cs[c]), len(cs[c]))
    rows = []
    for n, d in sorted(step.items()):
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f, w = d["FETCH_SIZE"][0], d["WRITE_SIZE"][0]
            rows.append({"kernel": n, "launches": d["FETCH_SIZE"][1], "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                         "traffic_bytes": (2 * f + w) * 1024})
    json.dump(rows, open(f"{OUT}/{tag}_pmc_step_kernels{sfx}.json", "w"), indent=1)
    print(f"---- {prec}")
    for r in rows:
        print(f"{r['kernel'][:70]:70s} n={r['launches']:3d} fetch {r['FETCH_SIZE_KB'] / 1024:8.1f} MiB write {r['WRITE_SIZE_KB'] / 1024:8.1f} MiB")
    hb = [r for r in rows if r["kernel"].startswith(head_kernel)]
    if hb:
        r = dict(hb[0])
        r["note"] = ("traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction for 16-byte-per-lane streams; an upper estimate "
                     "where loads are narrower); separate rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1 --no-graph, B=64 100x100")
        json.dump(r, open(f"{OUT}/{tag}_pmc_head_bwd{sfx}.json", "w"), indent=1)
    cb = [r for r in rows if r["kernel"].startswith("conv3x3_bwd_s3_kernel<8, false>")]
    if cb and prec == "fp32":
        r = dict(cb[0])
        r["alg_bytes_mean_launch"] = (3 * 2 * 64 * 128 * 128 * 24 * 4 + 2 * 64 * 64 * 64 * 24 * 4 + 2 * 64 * 64 * 64 * 40 * 4) / 5
        r["note"] = ("roofline.traffic of the fp32 line (round 6): mean over the step's five launches of the fused conv backward (three at 128 x 128 x 2 "
                     "streams, up2b at 64 x 64 x 2, the two skip halves of up2a at 64 x 64 x 4 problems); traffic = 2*FETCH_SIZE + WRITE_SIZE "
                     "(gfx950 FETCH_SIZE correction for 16-byte-per-lane streams); separate rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1 "
                     "--no-graph, B=64 100x100")
        json.dump(r, open(f"{OUT}/{tag}_pmc_conv_bwd.json", "w"), indent=1)
    conv = collections.defaultdict(dict)
    for d in glob.glob(f"{OUT}/pmcc_{prec}_*"):
        for n, cs in per_kernel(f"{d}/**/*counter_collection.csv").items():
            if n.startswith(conv_kernel):
                for c, v in cs.items():
                    conv[n][c] = mean(v)
    for n, d in conv.items():
        print(n, {k: round(v, 1) for k, v in d.items()})
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out = {"kernel": n, "counters": d, "traffic_bytes": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
                   "alg_bytes": 4 * 64 * 128 * 128 * esz * 16,
                   "note": "grouped conv 8->8 @128x128 x4 (tools/ablate_conv_group.py 8 8 128, ABL_ONE=0), mean per launch; traffic = "
                           "2*FETCH_SIZE + WRITE_SIZE; MFMA pipe utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): "
                           "both counters are summed over the chip (VERDICT round 5: the earlier note divided by the summed GRBM cycles)",
                   "mfma_busy_per_simd_over_kernel_cycles": (d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)
                                                              if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("GRBM_GUI_ACTIVE") else None),
                   "wait_inst_any_over_wave_cycles": (d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"] if d.get("SQ_WAVE_CYCLES") else None)}
            json.dump(out, open(f"{OUT}/{tag}_pmc_conv_8to8{sfx}.json", "w"), indent=1)
