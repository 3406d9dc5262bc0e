"""Print the figures DESIGN.md / README.md quote from one profile directory of tools/profile_all.sh:
    python tools/design_numbers.py gpurun_out/prof_r06_b"""
import csv, json, sys
O = sys.argv[1]
d = json.loads([l for l in open(O + "/bench.json") if l.startswith("{")][-1])
r = d["roofline"]
print("sha", d["kernel_sha"], "value", round(d["value"]), "dense", round(d["value_dense"]), "ms/step", round(d["ms_per_step"], 2), "views/s", round(d["views_per_s"]))
print("dominant frac", round(r["frac"], 4), "ms", round(r["launch_ms"], 4), "hbm_measured", round(r["hbm_measured"], 4))
for k, v in r["stages"].items():
    print(" ", k, "ms", v["ms"], "GBps", round(v["GBps"]), "frac", round(v["frac"], 4), "fused", round(v["frac_fused"], 4), "hbm", round(v.get("hbm_measured", 0), 4))
w = r["whole_view"]; print("whole ms", round(w["ms"], 4), "frac", round(w["frac"], 4), "fused", round(w["frac_fused"], 4), "hbm", round(w["hbm_measured"], 4))
c = r["convolve_noise"]; print("conv+noise ms", round(c["ms"], 4), "frac", round(c["frac"], 4), "with rotate", round(c["frac_with_rotate_kernel"], 4))
for k, v in r["passes"].items():
    if k != "planes": print(" ", k, v["ms"], round(v["GBps"]), round(v["frac"], 3))
print("stage_ms", r["stage_ms"])
print("serial", round(d["serial"]["value"]), round(d["serial"]["ms_per_step"], 2), "dense rot", d["no_empty_space"]["rotate_attenuate_ms"], "ext", d["no_empty_space"]["extract_ms"])
print("main_iteration", round(d["main_iteration"]["ms_per_iteration"], 3), round(d["main_iteration"]["value"]), "two_streams", round(d["two_streams"]["value"]))
e = d["end_to_end"]; print("e2e same", round(e["same_ground_truth"]["ms_per_view"], 2), "fresh", round(e["fresh_ground_truth_per_view"]["ms_per_view"], 2), "f32", round(e["same_ground_truth_float32_transfer"]["ms_per_view"], 2), e["acquisition_transfer"]["views_as_uint16"], e["acquisition_transfer"]["fell_back_to_float32"])
s = d["size_1024"]; print("1024 value", round(s["value"]), "ms", round(s["ms_per_view"], 2), "serial", round(s["serial"]["ms_per_view"], 2), "traffic", s["roofline"]["traffic"], "frac", round(s["roofline"]["frac"], 4), "hbm", round(s["roofline"]["hbm_measured"], 4))
for k, v in d["small_views"].items():
    if k != "note": print(" ", k, "seq", round(v["sequential_Mvoxel_per_s"]), "stacked", round(v["value"]), v["ms_per_view"])
cb = d["cpu_baseline"]; print("cpu", round(cb["value"], 2), round(cb["modes"]["all_cores"]["value"], 1), cb["cores"])
for row in csv.DictReader(open(O + "/kernel_stats.csv")):
    print(" ", row["Name"][:70], row["Calls"], round(float(row["AverageNs"]) / 1e3, 1))
print(open(O + "/small_views.txt").read())
print(open(O + "/other_sizes.txt").read()[:1200])
