#!/usr/bin/env python3
"""Turns the scratch output of a profiling call (gpurun_out/) into the committed, stamped measurements under profiles/.

  tools/stamp_profiles.py traffic <profile_dir> <tag>   tools/profile_bench.sh output -> profiles/<tag>_traffic.json,
                                                         <tag>_kernel_stats.csv, <tag>_rocprofv3_summary.json
  tools/stamp_profiles.py c3n1 <bench_json> <tag>       `bench.py --workload c3` line -> profiles/<tag>_c3_n1.json
  tools/stamp_profiles.py pmc <prof_dir> <name>         tools/prof_counters.sh output -> profiles/<name>.txt (stamped)
  tools/stamp_profiles.py round <round_dir> <tag>       tools/measure_round.sh output (gpurun_out/round_<tag>, with the C4
                                                         profile in gpurun_out/profile_<tag>_c4) -> profiles/<tag>_*

Every file carries `lib_hash` = the hash of the HIP library's sources (fastsmc_amd.build.hip_source_hash()); bench.py
reports a committed measurement only when that hash is the current one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsmc_amd.build import hip_source_hash  # noqa: E402


def traffic(src: str, tag: str) -> None:
    summ = json.load(open(os.path.join(src, "summary.json")))
    line = json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
    pmc = summ["pmc_per_launch"]
    fetch_kb, write_kb = pmc["FETCH_SIZE"]["mean"], pmc["WRITE_SIZE"]["mean"]
    stats = summ["kernel_stats_csv"]
    kern = [ln for ln in stats.splitlines() if "decode_kernel" in ln][0]
    name = kern.split('",')[0].strip('"')
    avg_ns = float(kern.split('",')[1].split(",")[2])
    wl = line["config"]["workload"]
    key = "c2:1000x50000:K69" if "1000 haplotypes x 50000" in wl else wl
    out = {"workload": wl, "workload_key": key, "beta_stride": line["config"]["beta_stride"], "kernel": name,
           "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "correction": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024.0, "kernel_avg_ns_rocprofv3": avg_ns,
           "kernel_ms_hip_events_same_run": line["roofline"]["kernel_ms"], "lib_hash": hip_source_hash()}
    assert line["config"]["lib_hash"] == out["lib_hash"], "the profile was taken with another build of the library"
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w").write(stats)
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_summary.json"), "w"), indent=1)
    json.dump(line, open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_kernel_trace.json"), "w"))
    print(json.dumps(out, indent=1))


def c3n1(src: str, tag: str) -> None:
    line = json.loads(open(src).read().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong"
    assert line["config"]["lib_hash"] == hip_source_hash(), "measured with another build of the library"
    wl = line["config"]["workload"]
    import re

    m = re.search(r"synthetic (\d+) haplotypes x (\d+) sites, K=(\d+), a seeded sub-list of (\d+)", wl)
    key = f"c3:{m.group(1)}x{m.group(2)}:K{m.group(3)}:{m.group(4)}"
    out = {"workload": wl, "workload_key": key, "value": line["value"], "unit": line["unit"],
           "ms_per_step": line["ms_per_step"], "kernel_ms": line["roofline"]["kernel_ms"],
           "roofline_frac": line["roofline"]["frac"], "lib_hash": hip_source_hash(), "bench_line": line}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_c3_n1.json"), "w"), indent=1)
    print(key, out["value"])


def round_files(src: str, tag: str) -> None:
    import shutil

    h = hip_source_hash()
    prof = os.path.join(ROOT, "profiles")
    c4 = os.path.join(os.path.dirname(os.path.abspath(src)), f"profile_{tag}_c4")
    summ = json.load(open(os.path.join(c4, "summary.json")))
    line = json.loads(open(os.path.join(c4, "bench_under_trace.json")).read().strip().splitlines()[-1])
    assert line["config"]["lib_hash"] == h, "the C4 profile was taken with another build of the library"
    stats = summ["kernel_stats_csv"]
    kern = [ln for ln in stats.splitlines() if "decode_kernel" in ln][0]
    avg_ns = float(kern.split('",')[1].split(",")[2])
    pmc = summ["pmc_per_launch"]
    hbm = (2 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    out = {"workload": line["config"]["workload"], "kernel": kern.split('",')[0].strip('"'),
           "kernel_avg_ns_rocprofv3": avg_ns, "kernel_ms_hip_events_same_run": line["roofline"]["kernel_ms"],
           "FETCH_SIZE_KB_per_launch": pmc["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB_per_launch": pmc["WRITE_SIZE"]["mean"],
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
           "frac_algorithmic_from_rocprof_avg": alg / (avg_ns * 1e-9) / 8e12,
           "frac_measured_bytes": hbm / (avg_ns * 1e-9) / 8e12, "lib_hash": h, "bench_line": line}
    json.dump(out, open(os.path.join(prof, f"{tag}_c4_at_size.json"), "w"), indent=1)
    open(os.path.join(prof, f"{tag}_c4_at_size_kernel_stats.csv"), "w").write(stats)
    for a, b in (("c4_reduced.json", "c4_reduced_600x3000.json"), ("family.jsonl", "family_members_c2_shape.jsonl"),
                 ("short.json", "short_windows.json"), ("short_kernel_stats.csv", "short_windows_kernel_stats.csv"),
                 ("run_c2.json", "run_c2_product_path.json"), ("bench_default.json", "bench_default.json"),
                 ("c1.json", "c1_shape.jsonl"), ("identify.jsonl", "identify.jsonl"),
                 ("identify_kernel_stats.csv", "identify_kernel_stats.csv"),
                 ("hashing.json", "hashing_c2_files_end_to_end.json"), ("sequence.jsonl", "sequence_mode.jsonl"),
                 ("wide_w2.jsonl", "wide_four_waves_per_group.jsonl"), ("c1_bench.json", "c1_bench_line.json"),
                 ("c5_job.json", "c5_job_window.json"), ("ingest_c3.json", "ingest_c3.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(prof, f"{tag}_{b}"))
    c3n1(os.path.join(src, "c3_n1.json"), tag)


def pmc(src: str, name: str) -> None:
    """tools/prof_counters.sh output (gpurun_out/prof_<tag>/summary.txt) -> profiles/<name>.txt; refused unless the
    summary carries the hash of the library as it is now."""
    txt = open(os.path.join(src, "summary.txt")).read()
    first = txt.splitlines()[0].split("\t")
    assert first[0] == "lib_hash" and first[1] == hip_source_hash(), "counters of another build of the library (or unstamped)"
    open(os.path.join(ROOT, "profiles", f"{name}.txt"), "w").write(txt)
    print(f"profiles/{name}.txt")


if __name__ == "__main__":
    {"traffic": traffic, "c3n1": c3n1, "round": round_files, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
