"""A/B timing of the hot kernels (F + J assembly, residual-only assembly, Jacobian SpMV) on the
bench mesh under different build/run-time switches; one child process per configuration (the
switches are read when the context is created).  Prints one JSON line per configuration.

usage: kernel_ab.py [mesh] -- runs the built-in list;  kernel_ab.py --one  (child)
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS_R1 = [
    dict(FEDM_ASSEMBLY_LEAN="1", FEDM_XCD_REMAP="0", FEDM_PATCH_ORDER="0"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="0", FEDM_PATCH_ORDER="0"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="0"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="16,16"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="32,32"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="16,32"),
    dict(FEDM_ASSEMBLY_LEAN="2", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="64,64"),
    dict(FEDM_ASSEMBLY_LEAN="1", FEDM_XCD_REMAP="1", FEDM_PATCH_ORDER="16,16"),
]


# round-2: micro-colouring with / without the gather-read criterion
CONFIGS = [
    dict(FEDM_PATCH_ORDER_READS="0"),
    dict(FEDM_PATCH_ORDER_READS="1"),
    dict(FEDM_PATCH_ORDER_READS="2"),
    dict(FEDM_PATCH_ORDER_READS="4"),
    dict(FEDM_PATCH_ORDER_READS="0"),
    dict(FEDM_PATCH_ORDER_READS="1"),
]


def child(n):
    from fedm_amd.cases import streamer
    msh = streamer.mesh(n, 4.0)
    prob = streamer.device_problem(msh.coords, msh.cells)
    streamer.initialise(prob, multigrid=False)
    prob.set_step(5e-12, 5e-12)
    out = {k: os.environ.get(k) for k in ("FEDM_ASSEMBLY_LEAN", "FEDM_XCD_REMAP", "FEDM_PATCH_ORDER", "FEDM_SKIP_CONST_PLANES", "FEDM_PATCH_ORDER_READS")
           if os.environ.get(k) is not None}
    for name, kind in (("FJ", 0), ("F", 2), ("spmv", 1)):
        prob.time_kernel(kind, 5)
        out[name + "_us"] = round(1e3 * min(prob.time_kernel(kind, 40) for _ in range(3)), 2)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        child(int(os.environ.get("FEDM_AB_MESH", "576")))
    else:
        mesh = sys.argv[1] if len(sys.argv) > 1 else "576"
        for cfg in CONFIGS:
            env = dict(os.environ, FEDM_AB_MESH=mesh, **cfg)
            r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True,
                               timeout=600)
            print(r.stdout.strip() or ("FAILED " + json.dumps(cfg) + r.stderr[-800:]), flush=True)
