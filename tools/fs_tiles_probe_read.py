"""Reads the kernel trace of tools/fs_tiles_probe.py: per configuration (in launch order) the durations of the
species-sweep kernels.  usage: python tools/fs_tiles_probe_read.py TRACE.csv [OUTPUT_OF_THE_PROBE]"""
import csv
import sys

rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1])))
# every fieldsplit_apply starts with species_planes_kernel
groups, cur = [], None
for s, e, n in rows:
    if 'species_planes_kernel' in n:
        cur = []
        groups.append(cur)
    elif cur is not None and ('fs_tile_sweeps' in n or 'fs_species_sweep' in n):
        cur.append((e - s) / 1e3)
labels = [l.split(":")[0].replace("CONFIG ", "") for l in open(sys.argv[2]) if l.startswith("CONFIG")] if len(sys.argv) > 2 else []
for i, g in enumerate(groups):
    label = labels[i // 2] if i // 2 < len(labels) else ""     # (every configuration is applied twice)
    print(f"{label:36s} {len(g)} launches, total {sum(g):7.1f} us:", " ".join(f"{d:.1f}" for d in g))
