"""A/B timing of the hot kernels (F + J assembly incl. its boundary kernel, residual-only assembly,
Jacobian SpMV: `fedm_time_kernel`, back to back -- the SpMV therefore runs from the Infinity Cache)
on the bench mesh under run-time switches; one child process per configuration (the switches are
read when the context is created).  Prints one JSON line per configuration.

usage: kernel_ab.py [mesh | -k] [KEY=VAL[,KEY=VAL...]] ...      one argument per configuration, e.g.
    kernel_ab.py 576 FEDM_PATCH_ORDER=0 FEDM_PATCH_ORDER=16,16 FEDM_SKIP_CONST_PLANES=0
Switches: FEDM_ASSEMBLY_LEAN=0|2|3, FEDM_XCD_REMAP=0|1, FEDM_PATCH_ORDER=group,mod|0,
FEDM_PATCH_ORDER_READS=w, FEDM_SKIP_CONST_PLANES=0|1, FEDM_SPMV_SKIP_ZERO_PLANES=0|1,
FEDM_HIP_LIB=<experiment build>.
Boxes of the pool differ by up to 12 % (MI355X_MICROARCH.md, DVFS): compare within one call only.
Round-2 results: DESIGN.md 8.1.
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KEYS = ("FEDM_ASSEMBLY_LEAN", "FEDM_XCD_REMAP", "FEDM_PATCH_ORDER", "FEDM_PATCH_ORDER_READS",
        "FEDM_SKIP_CONST_PLANES", "FEDM_SPMV_SKIP_ZERO_PLANES", "FEDM_HIP_LIB", "FEDM_PATCH_CLASSES",
        "FEDM_LEAN3_CLASSES", "FEDM_LEAN3_PERSISTENT", "FEDM_LEAN3_WGS_PER_CU")


def child(n):
    from fedm_amd.cases import streamer
    if n < 0:     # -k: the refined unstructured mesh with k um in the channel (bench.py's headline mesh: -4)
        h = -n * 1e-6
        msh = streamer.refined_mesh(h, growth=0.1, channel=(0.0, 100.0 * h) + streamer.CHANNEL[2:])
    else:
        msh = streamer.mesh(n, 4.0)
    prob = streamer.device_problem(msh.coords, msh.cells)
    streamer.initialise(prob, multigrid=False)
    prob.set_step(5e-12, 5e-12)
    out = {k: os.environ[k] for k in KEYS if k in os.environ}
    for name, kind in (("FJ", 0), ("F", 2), ("spmv", 1)):
        prob.time_kernel(kind, 5)
        out[name + "_us"] = round(1e3 * min(prob.time_kernel(kind, 40) for _ in range(3)), 2)
    print(json.dumps(out), flush=True)


def parse(arg):
    """'A=1,B=2,3' -> {'A': '1', 'B': '2,3'} (a comma starts a new key only when followed by KEY=)."""
    cfg, key = {}, None
    for part in arg.split(","):
        if "=" in part and part.split("=", 1)[0] in KEYS:
            key, val = part.split("=", 1)
            cfg[key] = val
        elif key is not None:
            cfg[key] += "," + part
    return cfg


if __name__ == "__main__":
    if "--one" in sys.argv:
        child(int(os.environ.get("FEDM_AB_MESH", "576")))
    else:
        args = sys.argv[1:]
        mesh = args.pop(0) if args and args[0].lstrip("-").isdigit() else "576"
        for cfg in ([parse(a) for a in args] or [{}]):
            env = dict(os.environ, FEDM_AB_MESH=mesh, **cfg)
            r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True,
                               timeout=600)
            print(r.stdout.strip() or ("FAILED " + json.dumps(cfg) + r.stderr[-800:]), flush=True)
