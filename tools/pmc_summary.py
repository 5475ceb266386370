#!/usr/bin/env python3
"""profiles/rNN_pmc_{fetch,write}_size.csv -> profiles/rNN_pmc_summary.json (HBM bytes per launch and kernel family).

FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(128-byte read requests are tallied as 64 bytes)."""
import collections, csv, json, os, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")

def family(name):
    n = re.sub(r"^void ", "", name).replace("ssdr::", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    # bench.py's ProfScope names: one per kernel template since round 6
    m = re.match(r"lfa32_l0_kernel", n)
    if m:
        return "lfa32_l0_kernel"
    m = re.match(r"lfa32_res_kernel<(\d+)", n)
    if m:
        return "lfa32_res_kernel<%s>" % m.group(1)
    m = re.match(r"lfa32_kernel<(\d+)", n)
    if m:
        return "lfa32_kernel<%s>" % m.group(1)
    m = re.match(r"dense_bf16_kernel<(\d+), ?(\d+)", n)
    if m:
        return "dense_bf16_kernel<%s,%s>" % (m.group(1), m.group(2))
    m = re.match(r"dense_chain_kernel<(\d+)", n)
    if m:
        return "dense_chain_kernel<%s>" % m.group(1)
    for fam in ("lfa_att_kernel", "lfa_bf16_kernel", "dense_rows", "dense_small_kernel", "dense_kernel", "tail_bf16_kernel", "tail_kernel"):
        if n.startswith(fam):
            return "dense_rows_kernel" if fam == "dense_rows" else ("dense_kernel" if fam.startswith("dense_") else ("lfa_att_kernel" if fam.startswith("lfa") else "tail_kernel"))
    # the profiler sites of round 5 (bench.py's roofline.others): one family per site
    for pre, fam in (("fe_scatter", "fe_scatter"), ("fe_reduce", "fe_reduce"), ("fe_row", "fe_rows_move"), ("fe_move", "fe_rows_move"), ("fe_", "fe_bbox_count_scan"),
                     ("tile_", "tile_select"), ("sel_chamfer_dir", "sel_chamfer"), ("fps_", "fps_chain"), ("kc_init", "fps_chain"), ("fill_double", "fps_chain"),
                     ("gather_max", "gather_max_kernel"), ("sel_region_stats", "sel_region_stats"), ("sel_rank", "sel_rank"), ("sel_class_hist", "sel_clsbal"), ("sel_clsbal", "sel_clsbal"),
                     ("cand_", "sel_candidate_rule"), ("sel_segment_mean", "sel_features_pack"), ("sel_centres", "sel_features_pack"), ("sel_chamfer_", "sel_features_pack"),
                     ("sel_adj", "sel_adjacency_propagate"), ("sel_propagate", "sel_adjacency_propagate"), ("kd_", "knn_tree_handover")):
        if n.startswith(pre):
            return fam
    m = re.match(r"grid_(search|retry)_kernel<(\d+)", n)
    if m:
        return "knn_grid_search<%s>" % m.group(2)
    if n.startswith("grid_"):
        return "knn_grid_build"
    m = re.match(r"kd_search_kernel<(\d+)", n)
    return "kd_search_kernel<%s>" % m.group(1) if m else n

def load(fn):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(D, fn))):
        a = acc[family(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024.0
    return acc

f, w = load("%s_pmc_fetch_size.csv" % R), load("%s_pmc_write_size.csv" % R)
import subprocess
try:
    commit = subprocess.check_output(["git", "-C", os.path.dirname(D), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "?"
MFMA = {}
try:      # optional third pass: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel family (tools/collect_profiles.sh)
    busy = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for r in csv.DictReader(open(os.path.join(D, "%s_pmc_mfma_busy.csv" % R))):
        b = busy[family(r["Kernel_Name"])]
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": b[0] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "SQ_BUSY_CYCLES": b[1] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE": b[2] += float(r["Counter_Value"])
    MFMA = {k: v for k, v in busy.items() if v[0] > 0}
except FileNotFoundError:
    pass
out = {"commit": commit, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                 "--no-pipeline` (tools/collect_profiles.sh); counters are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                 "128-B requests as 64 B), which over-counts kernels whose reads are mostly 64-B gathers", "kernels": {}}
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f[k][1] + w[k][1])):
    n = max(f[k][0], w[k][0], 1)
    out["kernels"][k] = {"launches": n, "fetch_size_bytes_per_launch_raw": int(f[k][1] / n), "fetch_size_bytes_per_launch_x2_gfx950": int(2 * f[k][1] / n),
                         "write_size_bytes_per_launch": int(w[k][1] / n), "hbm_bytes_per_launch": int((2 * f[k][1] + w[k][1]) / n)}
    if k in MFMA:      # SQ_VALU_MFMA_BUSY_CYCLES sums over the 4 SIMDs of every CU and over the XCDs' SQ_BUSY_CYCLES samples: report the raw sums and the
        out["kernels"][k]["sq_valu_mfma_busy_cycles"] = int(MFMA[k][0]); out["kernels"][k]["sq_busy_cycles"] = int(MFMA[k][1])      # ratio for relative comparison
        # utilisation of the matrix pipes: busy cycles (summed over the chip's 1024 SIMDs) / (1024 x the kernels' cycles); GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so the kernels' cycles are GRBM_GUI_ACTIVE / 8
        if MFMA[k][2] > 0:
            out["kernels"][k]["grbm_gui_active"] = int(MFMA[k][2])
            out["kernels"][k]["mfma_utilisation"] = round(MFMA[k][0] / (1024.0 * MFMA[k][2] / 8.0), 4)
# HBM bytes of ONE step: every kernel family's bytes per launch x its launches per step (a step = one selection: the FPS kernel runs once)
# steps in the pass = launches of the FPS kernel proper (one per selection), counted on the kernel's own name
steps = max(1, sum(1 for r in csv.DictReader(open(os.path.join(D, "%s_pmc_fetch_size.csv" % R)))
                   if re.sub(r"^void ", "", r["Kernel_Name"]).replace("ssdr::", "").replace("(anonymous namespace)::", "").startswith(("fps_block", "fps_coop", "fps_step"))))
out["steps_in_pass"] = steps
out["step_hbm_bytes"] = int(sum(out["kernels"][k]["hbm_bytes_per_launch"] * max(1, round(out["kernels"][k]["launches"] / steps)) for k in out["kernels"]
                                if not k.startswith("__amd_rocclr")))
json.dump(out, open(os.path.join(D, "%s_pmc_summary.json" % R), "w"), indent=1)
print("wrote", len(out["kernels"]), "kernel families")
