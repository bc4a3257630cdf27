"""Turns the CSVs of scratch/pmc_frame.sh (rocprofv3 --pmc passes over `bench.py --steps 3 --warmup 1 --no-ibl
--no-coverage --cpu-rows 0`) into profiles/rN_counters.json, the per-launch counter figures bench.py puts beside
its timings.  usage: python scratch/pmc_to_json.py gpurun_out/<dir> profiles/rN_counters.json [gpurun_out/<ibl dir> gather_bench.txt]
The optional pair adds the L2-miss line traffic of the irradiance and prefilter kernels (scratch/pmc_ibl2.sh) and the
random-line read rates of scratch/ubench/gather_bench."""
import collections, csv, glob, json, sys

root, out = sys.argv[1], sys.argv[2]
KERNELS = {"shade_kernel": "shade_kernel", "light_classes_kernel": "light_classes_kernel", "bloom_post_fused_kernel": "bloom_post_fused", "bloom_h_kernel": "bloom_h_kernel", "bloom_v_kernel": "bloom_v_kernel"}
acc = {k: collections.defaultdict(list) for k in KERNELS}
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            for k, pat in KERNELS.items():
                if pat in row["Kernel_Name"]:
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
W, H, L = 3840, 2160, 64
rec = {"_comment": "Per-launch means of rocprofv3 --pmc counters, one pass per counter group (scratch/pmc_frame.sh). traffic_bytes = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE and WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes.",
       "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-ibl --no-coverage --no-ranges-compare --cpu-rows 0",
       "workload": {"width": W, "rows": H, "lights": L, "cube_res": 1024, "dir_res": 4096}, "kernels": {}}
for k, ctr in acc.items():
    if not ctr:
        continue
    m = {n: sum(v) / len(v) for n, v in ctr.items()}
    e = {"launches_sampled": max(len(v) for v in ctr.values())}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e.update(fetch_size_kib=round(m["FETCH_SIZE"], 1), write_size_kib=round(m["WRITE_SIZE"], 1), traffic_bytes=int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024))
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE"):
        if n in m:
            e[n] = int(round(m[n]))
    rec["kernels"][k] = e
if len(sys.argv) > 4:
    import re
    names = ["irradiance_rows_kernel", "irradiance_texel_kernel", "prefilter_texel_kernel", "equirect_kernel", "footprint_tiles_kernel", "brdf_lut_kernel"]
    ibl = {k: collections.defaultdict(float) for k in names}
    for path in glob.glob(sys.argv[3] + "/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                for k in ibl:
                    if k in row["Kernel_Name"]:
                        ibl[k][row["Counter_Name"]] += float(row["Counter_Value"])  # summed over the kernel's launches of the one run
    rec["ibl"] = {"command": "rocprofv3 --pmc <counters> -- python3 scratch/ibl_one.py (one run of the precompute at the 2048^2 configuration: "
                             "irradiance in the default order, the prefilter in two launches, then the irradiance pass in the shader's order)", "kernels": {}}
    for k, tot in ibl.items():
        if "TCC_MISS_sum" not in tot:
            continue
        e = {"TCC_MISS_sum": int(tot["TCC_MISS_sum"]), "TCC_HIT_sum": int(tot["TCC_HIT_sum"]), "l2_miss_line_bytes": int(tot["TCC_MISS_sum"]) * 128,
             "fetch_bytes": int(tot["FETCH_SIZE"] * 2 * 1024), "write_bytes": int(tot.get("WRITE_SIZE", 0.0) * 1024)}
        for n in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "TA_TA_BUSY_sum", "TCP_TCC_READ_REQ_sum", "GRBM_GUI_ACTIVE"):
            if n in tot:
                e[n] = int(tot[n])
        if e.get("GRBM_GUI_ACTIVE"):
            cycles = e["GRBM_GUI_ACTIVE"] / 8
            if "SQ_ACTIVE_INST_VALU" in e: e["valu_busy_frac"] = round(e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles), 3)
            if "TA_TA_BUSY_sum" in e: e["texture_addresser_busy_frac"] = round(e["TA_TA_BUSY_sum"] / 256 / cycles, 3)
            if "SQ_THREAD_CYCLES_VALU" in e and e.get("SQ_ACTIVE_INST_VALU"): e["lanes_active_frac"] = round(e["SQ_THREAD_CYCLES_VALU"] / (64 * e["SQ_ACTIVE_INST_VALU"]), 3)
            if "TCP_TCC_READ_REQ_sum" in e: e["l2_to_l1_line_bytes"] = e["TCP_TCC_READ_REQ_sum"] * 128
        rec["ibl"]["kernels"][k] = e
    rates = {}
    for line in open(sys.argv[4]):
        m = re.match(r"random lines, (\S+ \S+) table\s+m0\s+[\d.]+ cyc \(([\d.]+) TB/s", line)
        if m:
            rates[m.group(1)] = float(m.group(2)) * 1000.0
    rec["random_line_read_GBps"] = {"by_table_size": rates, "source": "scratch/ubench/gather_bench: every lane of every gather reads its own random 128-byte line "
                                    "(32 waves per CU); what the path behind a missing L2 delivers, whatever the kernel"}
with open(out, "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec["kernels"], indent=1))
