"""Summaries of tools/collect_profiles.sh -> profiles/<tag>_* (tracked).  Also usable locally on the
merged gpurun_out/<tag>_prof directory: python tools/copy_profiles.py r02 gpurun_out/r02_prof"""
import collections
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
tag, out = sys.argv[1], Path(sys.argv[2])
prof = ROOT / "profiles"
prof.mkdir(exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(str(out / pattern), recursive=True))
    if not hits:
        raise SystemExit(f"missing {pattern} under {out}")
    return Path(hits[0])


shutil.copy(one("trace/**/t_kernel_stats.csv"), prof / f"{tag}_kernel_stats.csv")
line = [ln for ln in (out / "bench_under_rocprof.json").read_text().splitlines() if ln.startswith("{")][-1]
(prof / f"{tag}_bench_under_rocprof.json").write_text(json.dumps(json.loads(line), indent=1) + "\n")


def load(path, counter):
    per_dispatch, names = collections.defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = (r.get("Dispatch_Id"), r["Kernel_Name"])
        per_dispatch[key] += float(r["Counter_Value"])       # one row per XCD / instance
        names[key] = r["Kernel_Name"]
    acc = collections.defaultdict(list)
    for key, v in per_dispatch.items():
        acc[names[key]].append(v)
    return acc


short = lambda name: re.sub(r"^void ", "", name).split("(")[0]
import bench
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, per launch, MI355X, one block per workload "
               "(unstructured / tensor: python3 bench.py --steps 2 --warmup 0 --repeats 1 --preroll 0 --family <workload> "
               "--no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge; glow_discharge: "
               "python3 tools/gd_steps.py 4).  Both counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): "
               "FETCH_SIZE counts half the bytes of a coalesced stream -> read bytes = 2*FETCH_SIZE; WRITE_SIZE is exact.",
       "kernel_source_sha": bench.kernel_source_sha(), "workloads": {}}
for W in ("unstructured", "tensor", "glow_discharge"):
    try:
        fetch = load(one(f"{W}/fetch/**/*counter_collection.csv"), "FETCH_SIZE")
        write = load(one(f"{W}/write/**/*counter_collection.csv"), "WRITE_SIZE")
    except SystemExit as exc:
        print("skipped", W, exc)
        continue
    block = {"kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        if not f or not w or "fedm::" not in name:
            continue
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        block["kernels"][short(name)] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa,
                                         "launches_sampled": min(len(f), len(w)),
                                         "read_bytes_corrected": 2.0 * fa * 1024.0, "write_bytes": wa * 1024.0,
                                         "traffic_bytes_corrected": (2.0 * fa + wa) * 1024.0}
    res["workloads"][W] = block
(prof / f"{tag}_pmc_traffic.json").write_text(json.dumps(res, indent=1) + "\n")

text = [f"rocprofv3 --pmc (two passes, 8 SQ counters each) on `python3 bench.py --steps 2 --warmup 0 --repeats 1 --preroll 0 "
        f"--family <mesh> --no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge`, averages per "
        f"launch; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  kernel sources "
        f"{bench.kernel_source_sha()}", ""]
def sq_section(W, kern, title, text):
    """SQ counters of the kernels whose name contains `kern` on mesh family W, appended to `text`"""
    if True:
        rows = []
        for d in ("sq1", "sq2"):
            try:
                path = one(f"{W}/{d}/**/*counter_collection.csv")
            except SystemExit:
                continue
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(path)):
                if kern in r["Kernel_Name"]:
                    per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for cname, dd in sorted(per.items()):
                v = list(dd.values())
                rows.append((cname, sum(v) / len(v), len(v)))
        if not rows:
            return
        c = {n: v for n, v, _ in rows}
        text.append(title)
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            pct = lambda k: 100.0 * c.get(k, 0.0) / wc
            text.append(f"share of a wave's lifetime: parked at s_waitcnt/s_barrier {pct('SQ_WAIT_ANY'):.1f} %, issue-stalled "
                        f"{pct('SQ_WAIT_INST_ANY'):.1f} % (LDS {pct('SQ_WAIT_INST_LDS'):.1f} %), instruction in flight "
                        f"{pct('SQ_ACTIVE_INST_ANY'):.1f} % (VALU {pct('SQ_ACTIVE_INST_VALU'):.1f} %, LDS {pct('SQ_ACTIVE_INST_LDS'):.1f} %, "
                        f"scalar {pct('SQ_ACTIVE_INST_SCA'):.1f} %)")
        if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0.0:
            text.append(f"LDS bank conflicts: {100.0 * c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']:.1f} % of the LDS-array cycles")
        text += [f"{n:32s} {v:16.1f}  (n={k})" for n, v, k in rows]
        text.append("")


for W in ("unstructured", "tensor"):
    for kern in ("assemble_lean3", "residual_lean3"):
        sq_section(W, kern, f"== {kern} kernel on the {W} mesh", text)
(prof / f"{tag}_pmc_assembly_sq.txt").write_text("\n".join(text) + "\n")
# the other kernels of a time step, same passes (the headline mesh)
step = [text[0].replace("assembly", "step"), "The kernels of a time step besides the assembly, largest shares of the step first (profiles/*_step_sequence.txt).", ""]
for kern, what in (("fs_tile_sweeps", "species sweeps on tiles (three sweeps a launch)"), ("ell_spmv_kernel", "multigrid cycle: products, sweeps, composite levels"),
                   ("spmv_dots_kernel", "Jacobian product fused with the Krylov step's dot products"), ("species_planes", "field split set-up behind an assembly"),
                   ("dense_gemv", "coarsest level: dense inverse"), ("boundary_dirichlet", "boundary facets + Dirichlet rows")):
    sq_section("unstructured", kern, f"== {kern}: {what}", step)
(prof / f"{tag}_pmc_step_kernels_sq.txt").write_text("\n".join(step) + "\n")
for W, block in res["workloads"].items():
    top = sorted(block["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_corrected"])[:6]
    for k, v in top:
        print(f'{W:14s} {v["traffic_bytes_corrected"] / 1e6:10.2f} MB  n={v["launches_sampled"]:4d}  {k}')
print("\n".join(text[:8]))
