#!/usr/bin/env python3
"""Rewrites the generated table of DESIGN.md section 5 (between the BENCH markers) from a bench line:
   tools/fill_docs.py profiles/<tag>_bench_n1.json
so that the numbers the documents quote are the numbers of a kept bench run, not transcriptions."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
j = json.load(open(src))
k = j["kernel_ms"]; one = j["one_step_at_a_time"]; dk = j["dominant_kernel"]; rf = j["roofline"]; cb = j.get("cpu_baseline", {}); m1 = j.get("map1", {})
long_run = None
lp = src.replace("_bench_n1.json", "_bench_n1_steps128.json")
if lp != src and os.path.exists(lp):
    long_run = json.load(open(lp))
# the DRIVER's measurement of the previous round next to the builder's own: K2's fraction moves with the box (0.43-0.51), and the
# driver's number is the one that is judged
import glob
drv = "n/a"
prev = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")))
if prev:
    try:
        pj = json.load(open(prev[-1])).get("parsed") or {}
        drv = "`%s`: %.1f Gpix/s, %.1f ms per step; `roofline.frac` **%.3f** (K2 launch %.3f ms)" % (
            os.path.basename(prev[-1]), pj["value"] / 1e3, pj["ms_per_step"], pj["roofline"]["frac"], pj["roofline"]["avg_launch_ms"])
    except Exception as e:
        drv = "n/a (%r)" % e
rows = [
    ("source", "`%s` (`python3 bench.py`, kernel sources sha %s)" % (os.path.relpath(src, ROOT), j.get("source_sha"))),
    ("timed region", "%d steps in flight, %d-wave region stage, help %s, %d timed steps after %d warm-up steps" % (
        j["config"]["steps_in_flight"], j["config"]["region_waves_per_image"], "on" if j["config"].get("help_across_workgroups") else "off", j["steps"], j["warmup"])),
    ("**`value`**", "**%.1f Gpix/s, %.1f ms per step, %.2f M lines/s** (%d lines per step)" % (j["value"] / 1e3, j["ms_per_step"], j["lines_per_s"] / 1e6, j["lines_per_step"])),
    ("the same with 128 timed steps (the drain of the steps in flight amortised)", "**%.1f Gpix/s, %.1f ms per step, %.2f M lines/s** (`%s`)" % (
        long_run["value"] / 1e3, long_run["ms_per_step"], long_run["lines_per_s"] / 1e6, os.path.relpath(lp, ROOT))) if long_run else ("long run", "n/a"),
    ("one step at a time", "%.1f ms = %.1f Gpix/s; kernels K1 %.2f (incl. the lineIm clear) / K2 %.2f / K3 %.2f / K4 %.1f / K5 %.2f ms: front end %.2f ms" % (
        one["ms_per_step"], one["value"] / 1e3, k["gauss"], k["gradient"], k["sort"], k["region"], k["lines"], k["gauss"] + k["gradient"] + k["sort"] + k["lines"])),
    ("… with the cost history of the previous step (`lsd_set_cost_history`: same maps again)", "%.1f ms = %.1f Gpix/s" % (
        j["one_step_at_a_time_with_cost_history"]["ms_per_step"], j["one_step_at_a_time_with_cost_history"]["value"] / 1e3)) if j.get("one_step_at_a_time_with_cost_history") else ("cost history", "n/a"),
    ("the front end inside the timed region (event to event, waits for a CU included)", "K1 %.0f / K2 %.0f / K3 %.0f / K5 %.0f ms" % tuple(
        j["kernel_ms_in_timed_region"][x] for x in ("gauss", "gradient", "sort", "lines"))) if j.get("kernel_ms_in_timed_region") else ("front end in the timed region", "n/a"),
    ("`roofline` (K2)", "%.0f GB/s algorithmic = **%.3f** of 8 TB/s (launch %.3f ms); PMC traffic %s GB per launch; device copy in the same process %s GB/s" % (
        rf["achieved"], rf["frac"], rf["avg_launch_ms"], "%.2f" % (rf["traffic"] / 1e9) if rf.get("traffic") else "n/a", "%.0f" % rf["measured_copy_GBs"] if "measured_copy_GBs" in rf else "n/a")),
    ("... the driver's own line of the round before (other box, same kernel)", drv),
    ("K4 cycles per image", "alone (8 waves, the library's defaults): mean %.1f M, max %.1f M; timed region (%s): mean %.1f M, max %.1f M" % (
        dk["cycles_per_image"]["mean"] / 1e6, dk["cycles_per_image"]["max"] / 1e6, dk["timed_region"]["variant"], dk["timed_region"]["cycles_per_image"]["mean"] / 1e6, dk["timed_region"]["cycles_per_image"]["max"] / 1e6)),
    ("whole step vs HBM", "%.1f GB algorithmic per step -> %.0f GB/s = %.3f of peak (`roofline_pipeline`)" % (j["roofline_pipeline"]["algorithmic_bytes_per_step"] / 1e9, j["roofline_pipeline"]["achieved"], j["roofline_pipeline"]["frac"])),
]
if "single_image_latency_ms" in j:
    rows.append(("one 2048² image (device entry point)", "%.1f ms; the port on one host core %.1f ms (%.1f x)" % (j["single_image_latency_ms"], cb.get("single_image_ms", float("nan")), j.get("single_image_latency_vs_port", float("nan")))))
if m1:
    rows.append(("mapValue_map1 (608×480, 7 lines)", "one host call (`lsd_map_cache` + `lsd_run`, PCIe included) %.2f ms, the LSD call alone %.2f ms (port: %.2f ms, %.1f x); 512 copies resident: %.2f ms = **%.2f M lines/s** = %.0f x the reference's 1 346 lines/s (cross-host), %.0f x the port on one core of this box, %.1f x the port on its %d usable cores" % (
        m1["single_call_ms"], m1["single_call_lsd_only_ms"], cb.get("map1", {}).get("ms", float("nan")), m1.get("single_call_vs_port", float("nan")), m1["batch512_ms"], m1["batch512_lines_per_s"] / 1e6,
        m1["lines_per_s_vs_reference_cross_host"], m1.get("vs_port_one_core", float("nan")), m1.get("vs_port_all_cores", float("nan")), cb.get("all_cores", {}).get("cores", 0))))
if cb:
    ac = cb.get("all_cores", {})
    rows.append(("`cpu_baseline` (port)", "one pinned core %.1f Mpix/s (%.1f k lines/s); %s cores %.0f Mpix/s, per core %.2f of one core alone; GPU / port: %.0f x one core, %.1f x all usable cores" % (
        cb["value"], cb["lines_per_s"] / 1e3, ac.get("cores", "?"), ac.get("value", float("nan")), ac.get("per_core_vs_one_core", float("nan")), j.get("vs_port_one_core", float("nan")), j.get("vs_port_all_cores", float("nan")))))
rm = j.get("real_maps")
if rm:
    rows.append(("the reference's own maps, un-tiled, 512 copies each (`real_maps`; ms / Mpix/s / x one core of the port / answers of certified sets)",
                 "; ".join("%s %.1f / %.0f k / %.0f x / %d" % (k, v["ms"], v["Mpix_per_s"] / 1e3, v.get("vs_port_one_core", float("nan")), v["set_answers"]) for k, v in rm.items())))
if "libm_sensitive_images" in j:
    rows.append(("`lsd_last_sensitivity` on the batch", "%d of %d images have decisions within the libm's noise (%d such decisions, speculative evaluations included)" % (
        j["libm_sensitive_images"], j["config"]["images_rank0"], j["libm_near_ties"])))
wb = j.get("writeback_map")
if wb and "step_ms_with" in wb:
    rows.append(("the in-place remap of the caller's maps (`LSD_FLAG_WRITEBACK_MAP`, not in the timed region)", "K1's window %.2f -> %.2f ms one step at a time%s" % (
        wb["gauss_ms_without"], wb["gauss_ms_with"], "; the timed configuration with it, on a private copy per step in flight restored by a device copy before every step: %.1f ms per step" % (
            wb["timed_configuration_ms_per_step_with_writeback_and_restore_copy"]) if "timed_configuration_ms_per_step_with_writeback_and_restore_copy" in wb else "")))
p = j.get("strong_scaling_projection", {}).get("gpus")
if p:
    lat = " / ".join("%s: %.1f ms (%.2f x)" % (g, p[g]["max_shard_ms"], p[g]["speedup"]) for g in ("1", "2", "4", "8"))
    rows.append(("strong-scaling projection, one step (slowest shard alone on this GPU)", lat))
    if "pipelined_ms_per_step" in p["1"] and all("pipelined_speedup" in p[g] for g in ("2", "4", "8")):
        thr = "1: %.1f ms per step (%d in flight)" % (p["1"]["pipelined_ms_per_step"], p["1"].get("pipelined_steps_in_flight", 8)) + " / " + " / ".join(
            "%s: slowest shard %.1f ms, fastest %.1f (%.2f x; %d in flight)" % (g, p[g]["pipelined_max_shard_ms_per_step"], p[g]["pipelined_min_shard_ms_per_step"], p[g]["pipelined_speedup"],
                                                                              p[g].get("pipelined_steps_in_flight", 8)) for g in ("2", "4", "8"))
        rows.append(("… in throughput mode (every rank keeps the same number of images in flight: 8 × N steps, at most 32)", thr))
    if all("balanced_speedup" in p[g] for g in ("2", "4", "8")):
        rows.append(("… with the cost-aware deal (`lsd_shard_balanced`)", "one step: " + " / ".join("%s: %.1f ms (%.2f x)" % (g, p[g]["balanced_max_shard_ms"], p[g]["balanced_speedup"]) for g in ("2", "4", "8")) +
                     ("; throughput mode: " + " / ".join("%s: %.1f ms (%.2f x)" % (g, p[g]["pipelined_balanced_max_shard_ms_per_step"], p[g]["pipelined_balanced_speedup"]) for g in ("2", "4", "8")) if all("pipelined_balanced_speedup" in p[g] for g in ("2", "4", "8")) else "")))
table = "| | |\n|---|---|\n" + "\n".join("| %s | %s |" % r for r in rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a, b = s.index("<!-- BENCH:BEGIN -->"), s.index("<!-- BENCH:END -->")
s = s[:a] + "<!-- BENCH:BEGIN -->\n" + table + "\n" + s[b:]
open(path, "w").write(s)
print(table)
