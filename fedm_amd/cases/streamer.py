"""Positive streamer in air (Bagheri et al. 2018) on the device path.

Counterpart of examples/streamer_discharge/fedm-streamer.py: same deck
(file_input/benchmark_model), same boundary list, boundary-condition table,
initial conditions, Dirichlet values and solver settings.  The reference's
mesh.xml is not in its checkout, so the mesh is ours (tensor-product, optional
geometric grading towards the axis).
"""
from pathlib import Path

import numpy as np

from ..device import DeviceProblem, Model, Reaction
from ..mesh import Marking_boundaries, Mesh, RectangleMesh, geometric_lines
from ..termsum import TermSum, parse

U_W = 18750.0                 # fedm-streamer.py:39
BOX = 0.0125                  # :96-97
BOUNDARIES = [["line", 0.0, 0.0, 0.0, BOX], ["line", BOX, BOX, 0.0, BOX],
              ["line", 0.0, BOX, 0.0, 0.0], ["line", 0.0, BOX, BOX, BOX]]          # :98-101
BC_TYPE = [["zero flux", "Neumann"], ["zero flux", "Neumann"],
           ["zero flux", "zero flux"], ["zero flux", "zero flux"]]                 # :103-107
DECK = Path(__file__).resolve().parents[2] / "decks" / "streamer_discharge" / "file_input"
# deck strings, transport_coefficients/{e_Nb,e_ND,alpha}.dat:12 (used when no deck dir is given)
MU_E = "2.3987*E_m**(-0.26)"
D_E = "4.3628e-3*E_m**(0.22)"
ALPHA = "(1.1944e6 + 4.3666e26 * E_m**(-3))*exp(-2.73e7/E_m)-340.75"


def model(mu_e=MU_E, D_e=D_E, alpha=ALPHA, quadrature_degree=2):
    """LFA model of fedm-streamer.py:235-271."""
    mu = parse(mu_e)
    rate = parse(alpha) * mu * TermSum.field()          # alpha*mu[1]*E_m, :244-245
    return Model(n_species=2, poisson=True,
                 eq_type=["reaction", "drift-diffusion-reaction"], Z=[1.0, -1.0],
                 mu=[TermSum.const(0.0), mu], D=[TermSum.const(0.0), parse(D_e)],
                 reactions=[Reaction(rate, power=[0, 1], net=[1, 1])],
                 bc_kind=BC_TYPE, quadrature_degree=quadrature_degree)


def model_from_deck(file_input=DECK, model_name="benchmark_model", quadrature_degree=2):
    """The same model configured the way the script does it (fedm-streamer.py:47-53,
    227-245): species list, particle properties and transport coefficients through the
    deck readers; the 'fun:E' strings are parsed, never eval'd."""
    from .. import file_io, functions
    file_io.files.file_input = Path(file_input)
    path = file_io.files.file_input / model_name
    n_species, species, prop_files, _ = file_io.read_speclist(path)
    M, sign = file_io.read_particle_properties(prop_files, model_name)
    n_species, n_eq, species, M, sign = functions.modify_approximation_vars(
        "LFA", n_species, species, M, sign)
    _, D_y, _ = file_io.read_transport_coefficients(species, "Diffusion", model_name)
    _, mu_y, _ = file_io.read_transport_coefficients(species, "mobility", model_name)
    alpha = file_io.read_single_string(path / "transport_coefficients" / "alpha.dat")
    mu = [TermSum.coerce(v) for v in mu_y]
    D = [TermSum.coerce(v) for v in D_y]
    rate = parse(alpha) * mu[1] * TermSum.field()
    return Model(n_species=n_species, poisson=True,
                 eq_type=["reaction", "drift-diffusion-reaction"], Z=sign, mu=mu, D=D,
                 reactions=[Reaction(rate, power=[0, 1], net=[1, 1])],
                 bc_kind=BC_TYPE, quadrature_degree=quadrature_degree)


def mesh(n, grading=1.0):
    """n x n "right" mesh of the 1.25 cm box; grading>1 refines towards the axis."""
    x_lines = geometric_lines(BOX, n, grading) if grading != 1.0 else None
    return RectangleMesh((0.0, 0.0), (BOX, BOX), n, n, "right", x_lines=x_lines)


# the channel the streamer of this case travels in: from the seed at z = 1 cm towards the cathode,
# radius about 1 mm when it arrives (Bagheri et al. 2018, case 1)
CHANNEL = (0.0, 0.9e-3, 1.0e-3, 10.8e-3)           # (r0, r1, z0, z1) of the finest region


def refined_mesh(h_fine, growth=0.2, channel=CHANNEL, h_max=BOX / 12, n_levels=None, xml_path=None, retriangulate=True):
    """Locally refined unstructured mesh of the box: spacing ``h_fine`` in the streamer channel,
    growing with the distance from it (`fedm_amd.meshgen`): the stand-in for the reference's
    ``mesh.xml`` (fedm-streamer.py:116; missing from its checkout).  With ``xml_path`` the mesh
    is written as legacy DOLFIN XML and READ BACK through the reader a user mesh takes, so what
    the device gets went the way of ``Mesh('mesh.xml')``."""
    from .. import meshgen, mesh_io
    if n_levels is None:
        n_levels = max(2, int(np.ceil(np.log2(h_max / h_fine))) + 1)
    size = meshgen.box_distance_size(channel, h_fine, growth, h_max)
    msh = meshgen.refined_rectangle(BOX, BOX, size, h_fine, n_levels=n_levels, retriangulate=retriangulate)
    if xml_path is not None:
        mesh_io.write_dolfin_xml(msh, xml_path)
        msh = mesh_io.read_dolfin_xml(xml_path)
    return msh


def dirichlet(coords):
    """Phi = 0 at z = 0 and Phi = U_w at z = box height (fedm-streamer.py:186-200,233)."""
    z = coords[:, 1]
    cathode = np.nonzero(np.abs(z) < 3e-16)[0]
    anode = np.nonzero(np.abs(z - BOX) < 3e-16)[0]
    dofs = np.concatenate([cathode, anode]) * 3 + 2
    vals = np.concatenate([np.zeros(cathode.size), np.full(anode.size, U_W)])
    return dofs.astype(np.int32), vals


def initial_log_densities(coords):
    """fedm-streamer.py:169-172."""
    r, z = coords[:, 0], coords[:, 1]
    u_ion = np.log(1e13 + 5e18 * np.exp(-(r ** 2 + (z - 1e-2) ** 2) / (0.4e-3) ** 2))
    return u_ion, np.full_like(u_ion, np.log(1e13))


def device_problem(coords, cells, device=0, **model_kw):
    msh = Mesh(coords, cells)
    tags = Marking_boundaries(msh, BOUNDARIES)
    dofs, vals = dirichlet(msh.coords)
    return DeviceProblem(msh.coords, msh.cells, model(**model_kw), facet_tags=tags,
                         dirichlet_dofs=dofs, dirichlet_vals=vals, device=device)


# V(1,1): as effective as V(2,2) here at 75 % of the cost; Jacobi damping 0.85 instead of the
# default 2/3: the same 6 Krylov steps per time step early in the run, 35 instead of 38 later
# (tools/omega_sweep.py; 0.95: 40)
# V(1,1) with damped Jacobi; next to it a cycle with two Chebyshev sweeps per leg for the Newton
# solves that need many Krylov steps (once the streamer has formed: 27 instead of 37 Krylov steps
# per time step for 29 us more per cycle, tools/poly_cycle.py)
MULTIGRID = dict(nu=1, omega=0.85, hard_poly_degree=2)


def initialise(prob, multigrid=True):
    """Initial densities + the initial Poisson solve (fedm-streamer.py:169-225) on the device."""
    U = np.zeros((prob.nv, 3))
    U[:, 0], U[:, 1] = initial_log_densities(prob.coords)
    prob.set_state(U, U, U)
    if multigrid:
        prob.setup_multigrid(**MULTIGRID)
        # species block ~ diagonally scaled P1 mass matrix (spectrum in [0.5, 2]): a Chebyshev
        # polynomial in Duu^-1 Juu instead of plain block Jacobi.  Degree 6 makes a Newton system
        # cost 2 Krylov steps early in the run (degree 4: 3, block Jacobi: 5); later the potential
        # block limits the convergence (8-9 steps whatever the degree) and degree 4 is cheaper:
        # the library switches on the iteration count (tools/fs_sweeps.py).  (The potential-first
        # order of the split halves the late count but not the error: tools/fs_order_accuracy.py.)
        from ..device import chebyshev_weights
        prob.set_fieldsplit(chebyshev_weights(6), hard_weights=chebyshev_weights(4))
    its = prob.poisson_solve(rtol=1e-12)
    U = prob.get_state()
    prob.set_state(U, U, U)
    return U, its



class Stepper:
    """The script-level time loop of fedm-streamer.py:304-340 around one device problem."""

    partition_name = "single GPU"

    @property
    def assembly_kernel_name(self):
        sz = self.prob.sizes()
        if sz["assembly_variant"] == "lds-patches/one-pass":
            return (f"assemble_lean3_kernel<2,1,{sz['patch_threads']},kept planes> (LDS patches with micro-coloured cell "
                    "order, one pass over the cells, live planes only, F+J)")
        if sz["assembly_variant"] != "lds-patches":
            return f"volume assembly, variant '{sz['assembly_variant']}' (F+J)"
        return (f"assemble_lean2_kernel<2,1,{sz['patch_threads']}> (LDS patches with micro-coloured cell order, one "
                "equation row at a time, F+J)")

    @property
    def multigrid_levels(self):
        return getattr(self.prob, "multigrid_levels", None)

    def __init__(self, prob, dt_init=5e-12, dt_max=5e-12, dt_min=1e-15, ttol=1e-3,
                 relative_tolerance=1e-4, maximum_iterations=20, error_file=None, quiet=True):
        import tempfile
        from .. import functions as ff
        from ..forms import DeviceState, Expression, FunctionAssigner
        self.ff, self.prob = ff, prob
        self.solver = ff.PETScSNESSolver()
        self.solver.parameters["relative_tolerance"] = relative_tolerance
        self.solver.parameters["maximum_iterations"] = maximum_iterations
        self.problem = ff.Problem(None, None, [], device_problem=prob)
        self.dt = Expression("time_step", time_step=dt_init, degree=0)
        self.dt_old = Expression("time_step", time_step=1e30, degree=0)
        self.u_new, self.u_old = DeviceState(prob, "new"), DeviceState(prob, "old")
        self.assigner = FunctionAssigner()
        self.error, self.max_error = [0.0] * 2, [1] * 3
        self.ttol, self.dt_min, self.dt_max = ttol, dt_min, dt_max
        if error_file is None:
            error_file = Path(tempfile.mkdtemp(prefix="fedm_amd_")) / "relative error.log"
        self.error_file = Path(error_file)
        open(self.error_file, "w").close()
        self.t, self.steps, self.newton_iterations, self.linear_iterations = 0.0, 0, 0, 0
        self.quiet = quiet
        self.total_dofs = prob.n

    def initialise(self):
        return initialise(self.prob)

    def step(self):
        import contextlib, io
        ff = self.ff
        self.prob.shift_state()                              # :306-307
        sink = io.StringIO() if self.quiet else None
        with (contextlib.redirect_stdout(sink) if sink else contextlib.nullcontext()):
            self.t = ff.adaptive_solver(self.solver, self.problem, self.t, self.dt, self.dt_old,
                                        self.u_new, self.u_old, None, None, self.assigner,
                                        self.error, self.error_file, self.max_error, self.ttol,
                                        self.dt_min, time_dependent_arguments=[],
                                        approximation="LFA")
        self.newton_iterations += self.prob.last_report.iterations
        self.linear_iterations += self.prob.last_report.linear_iterations
        self.dt_old.time_step = self.dt.time_step            # :335
        self.dt.time_step = ff.adaptive_timestep(self.dt.time_step, self.max_error, self.ttol,
                                                 self.dt_min, self.dt_max)
        self.max_error[2] = self.max_error[1]
        self.max_error[1] = self.max_error[0]
        self.steps += 1
        return self.t

    def snapshot(self):
        """Checkpoint of the time loop: the three states stay on the device (fedm_state_snapshot), the
        script-level scalars (time, step sizes, error history, counters) are returned."""
        self.prob.snapshot_state()
        return dict(t=self.t, steps=self.steps, dt=self.dt.time_step, dt_old=self.dt_old.time_step,
                    error=list(self.error), max_error=list(self.max_error),
                    newton=self.newton_iterations, linear=self.linear_iterations)

    def restore(self, snap):
        self.prob.restore_state()
        self.t, self.steps = snap["t"], snap["steps"]
        self.dt.time_step, self.dt_old.time_step = snap["dt"], snap["dt_old"]
        self.error[:], self.max_error[:] = snap["error"], snap["max_error"]
        self.newton_iterations, self.linear_iterations = snap["newton"], snap["linear"]

    def log_rows(self):
        return [tuple(float(v) for v in line.split()) for line in open(self.error_file)]

    def sizes(self):
        return self.prob.sizes()

    def time_kernel(self, kind, repeats):
        return self.prob.time_kernel(kind, repeats)

    def profile(self, enable=True):
        self.prob.profile(enable)

    def profile_read(self):
        return self.prob.profile_read()


def run(prob, T_final=1e-10, max_steps=None, initialise_state=True, **kw):
    """Time loop of fedm-streamer.py:304-340 on the device.  Returns the error-log rows."""
    st = Stepper(prob, **kw)
    if initialise_state:
        st.initialise()
    while abs(st.t - T_final) / T_final > 1e-6:
        st.step()
        if max_steps is not None and st.steps >= max_steps:
            break
    return dict(log=st.log_rows(), t=st.t, steps=st.steps,
                newton_iterations=st.newton_iterations, linear_iterations=st.linear_iterations)
