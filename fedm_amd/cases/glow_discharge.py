"""Glow discharge in argon (LMEA, 4 particles) on the device path.

Counterpart of examples/glow_discharge/fedm-gd.py and its test harness
tests/integrated_tests/glow_discharge/fedm_gd.py: deck through ``fedm_amd.file_io``,
coefficient tables through ``Transport_/Rate_coefficient_interpolation`` and
``semi_implicit`` derivative tables (host numpy, once per time step), the Newton solves
(assembly with the nodal, semi-implicitly linearised coefficients, 'flux source' walls,
GMRES + field split) on the device.
"""
from pathlib import Path

import numpy as np

from .. import file_io, functions as ff
from ..device import DeviceProblem, GdModel
from ..forms import DeviceState, Expression, Function, FunctionAssigner
from ..mesh import Marking_boundaries, RectangleMesh
from ..physical_constants import elementary_charge, kB

DECK = Path(__file__).resolve().parents[2] / "decks" / "glow_discharge" / "file_input"


class _Nodal(Function):
    def __init__(self, n, value=0.0):
        super().__init__(values=np.full(n, float(value)))


class Case:
    """Set-up of fedm_gd.py:45-418 and one ``step()`` per pass of its time loop (:420-471)."""

    def __init__(self, nx=100, ny=100, file_input=DECK, device=0, T_final=1e-11,
                 relative_tolerance=1e-4, maximum_iterations=20, quiet=True, error_file=None,
                 device_pipeline=True, mesh=None, energy_loss=None, energy_Ei=0.0):
        import tempfile
        self.quiet = quiet
        model = "4_particles"
        file_io.files.file_input = Path(file_input)
        path = file_io.files.file_input / model
        self.Tgas, p0, self.U_w = 300.0, 1.0, -250.0
        self.N0 = N0 = p0 * 3.21877e22
        ns, species, prop, names = file_io.read_speclist(path)
        M, sign = file_io.read_particle_properties(prop, model)
        eq_type = ["reaction", "diffusion-reaction", "drift-diffusion-reaction", "drift-diffusion-reaction"]
        species_type = ["Neutral", "Neutral", "Ion", "electrons"]
        grad_diff = [t == "electrons" for t in species_type]
        P, L, G = file_io.reaction_matrices(path, species)
        kfiles = file_io.rate_coefficient_file_names(path)
        # (energy_loss: other losses than the deck's, e.g. with its sentinel values 7.77e77 / 9.99e99 for reactions
        # that lose Ei - mean energy / the mean energy, fedm/functions.py:906-909 -- the deck itself has none)
        self.energy_loss = file_io.read_energy_loss(path) if energy_loss is None else list(energy_loss)
        sentinels = any(7e77 < v < 8e77 or 9e99 < v < 1e100 for v in self.energy_loss)
        self.mu_x, self.mu_y, self.mu_dep = file_io.read_transport_coefficients(names, "mobility", model)
        self.D_x, self.D_y, self.D_dep = file_io.read_transport_coefficients(names, "Diffusion", model)
        self.k_dep = file_io.read_dependences(kfiles)
        self.k_x, self.k_y = file_io.read_rate_coefficients(kfiles, self.k_dep)
        self.De_diff = np.gradient(self.D_y[ns - 1], self.D_x[ns - 1]) / N0
        self.mue_diff = np.gradient(self.mu_y[ns - 1], self.mu_x[ns - 1]) / N0
        self.k_diff = [np.gradient(ky, kx) if dep == "Umean" else 0.0
                       for kx, ky, dep in zip(self.k_x, self.k_y, self.k_dep)]
        ns, n_eq, species, M, sign = ff.modify_approximation_vars("LMEA", ns, species, M, sign)
        self.ns, self.nr = ns, len(kfiles)

        self.gap = self.wall = 0.01
        # (mesh: any triangle mesh of the 1 cm x 1 cm domain instead of the script's crossed one)
        self.mesh = mesh = mesh if mesh is not None else RectangleMesh((0.0, 0.0), (self.wall, self.gap), nx, ny,
                                                                      "crossed")
        nv = mesh.num_vertices()
        boundaries = [["line", 0.0, 0.0, 0.0, self.wall], ["line", self.gap, self.gap, 0.0, self.wall],
                      ["line", 0.0, self.gap, 0.0, 0.0], ["line", 0.0, self.gap, self.wall, self.wall]]
        tags = Marking_boundaries(mesh, boundaries)
        ref_met, ref_zero = [0.3, 0.3, 5e-4, 0.3], [1.0] * 4
        vth = [0.0] * ns
        for i in range(1, ns - 1):
            vth[i] = np.sqrt(8.0 * kB * self.Tgas / (np.pi * M[i]))
        gd_model = GdModel(n_species=ns, N0=N0, eq_type=eq_type, grad_diffusion=grad_diff,
                           is_ion=[t == "Ion" for t in species_type], sign=sign, vth=vth,
                           electron_mass=M[ns - 1], power=P.tolist(), net=(G - L).tolist(),
                           energy_loss=self.energy_loss, energy_Ei=energy_Ei,
                           mean_energy_form="unknown_ratio" if sentinels else None,     # u[0] / u[n - 1], fedm_gd.py:365
                           ref=[ref_met, ref_met, ref_zero, ref_zero],
                           gamma=[0.06, 0.06, 0.0, 0.0], we_secondary=5.0, quadrature_degree=4)
        z = mesh.coords[:, 1]
        self.powered = np.nonzero(np.abs(z) <= 3e-16)[0]
        self.grounded = np.nonzero(np.abs(z - self.gap) <= 3e-16)[0]
        ddofs = np.concatenate([self.powered, self.grounded]) * n_eq + (n_eq - 1)
        self.prob = DeviceProblem(mesh.coords, mesh.cells, gd_model, facet_tags=tags,
                                  dirichlet_dofs=ddofs, dirichlet_vals=self.dirichlet_values(0.0),
                                  device=device)
        # initial conditions, fedm_gd.py:238-262 (equal charges, zero voltage at t=0 -> Phi = 0)
        n_ic = [N0, 1e12, 1e12, 1e12]
        U = np.zeros((nv, n_eq))
        U[:, 0] = np.log(3.0) + np.log(n_ic[ns - 1])
        for i in range(1, ns):
            U[:, i] = np.log(n_ic[i])
        self.prob.set_state(U, U, np.zeros_like(U))
        self.U = U
        self.mean_energy = _Nodal(nv, 3.0)
        self.mean_energy_old = _Nodal(nv, 3.0)
        self.redE = _Nodal(nv, 0.0)
        self.mu = [_Nodal(nv) for _ in range(ns)]
        self.D = [_Nodal(nv) for _ in range(ns)]
        self.k = [_Nodal(nv) for _ in range(self.nr)]
        self._mass = None
        ff.Transport_coefficient_interpolation("initial", self.mu_dep, N0, self.Tgas, self.mu, self.mu_x,
                                               self.mu_y, self.mean_energy, self.redE, self.mu)
        ff.Transport_coefficient_interpolation("initial", self.D_dep, N0, self.Tgas, self.D, self.D_x,
                                               self.D_y, self.mean_energy, self.redE, self.mu)
        ff.Rate_coefficient_interpolation("initial", self.k_dep, self.k, self.k_x, self.k_y,
                                          self.mean_energy, self.redE, Te=0, Tgas=0)
        self.prob.setup_multigrid(nu=1)
        # species block (energy + densities): a Chebyshev polynomial in Duu^-1 Juu instead of plain
        # block Jacobi -- degree 4 on [0.5, 2]: 51 -> 23 GMRES iterations per step at 200k DOFs
        # (tools/gd_sweeps.py); degree 8 on [0.3, 2.2] (the energy row widens the spectrum): another
        # third fewer, +14 % steps/s at 402k DOFs (tools/gd_cycle.py; the potential block is not what
        # limits these systems: the polynomial-smoother cycle changes nothing here)
        from ..device import chebyshev_weights
        self.prob.set_fieldsplit(chebyshev_weights(8, 0.3, 2.2))
        self.device_pipeline = device_pipeline
        if device_pipeline:
            self.upload_fields()               # 'initial' values incl. the const rows
            self._install_device_pipeline()
        # solver, time stepping (fedm_gd.py:104-133, 405-418)
        self.solver = ff.PETScSNESSolver()
        self.solver.parameters["relative_tolerance"] = relative_tolerance
        self.solver.parameters["maximum_iterations"] = maximum_iterations
        self.problem = ff.Problem(None, None, [], device_problem=self.prob)
        self.Phi_powered = Expression("U0*(1-exp(-t/1e-9))", U0=self.U_w, t=0.0, degree=0)
        self.problem.before_solve = lambda: self.prob.set_dirichlet_values(
            self.dirichlet_values(self.Phi_powered.t))
        self.dt = Expression("time_step", time_step=1e-13, degree=0)
        self.dt_old = Expression("time_step", time_step=1e30, degree=0)
        self.ttol, self.dt_min, self.dt_max, self.T_final = 2e-3, 1e-15, 1e-8, T_final
        self.error, self.max_error = [0.0] * (ns + 1), [1] * 3
        self.u_new, self.u_old = DeviceState(self.prob, "new"), DeviceState(self.prob, "old")
        self.assigner = FunctionAssigner()
        if error_file is None:
            error_file = Path(tempfile.mkdtemp(prefix="fedm_amd_gd_")) / "relative error.log"
        self.error_file = Path(error_file)
        open(self.error_file, "w").close()
        self.t, self.t_output, self.snapshot = 0.0, 1e-11, None
        self.newton_iterations = self.linear_iterations = 0

    def dirichlet_values(self, t):
        return np.concatenate([np.full(self.powered.size, self.U_w * (1.0 - np.exp(-t / 1e-9))),
                               np.zeros(self.grounded.size)])

    def project_reduced_field(self, Phi):
        """redE = project(1e21*sqrt(dot(-grad(Phi), -grad(Phi)))/N0), fedm_gd.py:432: consistent
        P1 mass solve of a cell-wise constant (host, post-processing size)."""
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        m = self.mesh
        x = m.coords[m.cells]
        d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
        if self._mass is None:
            vals = np.abs(det)[:, None, None] * ((np.ones((3, 3)) + np.eye(3)) / 24.0)[None]
            c = m.cells.astype(np.int64)
            rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
            cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
            n = m.num_vertices()
            self._mass = spla.splu(sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(n, n)).tocsc())
        P = Phi[m.cells]
        gx = (P[:, 0] * (x[:, 1, 1] - x[:, 2, 1]) + P[:, 1] * (x[:, 2, 1] - x[:, 0, 1])
              + P[:, 2] * (x[:, 0, 1] - x[:, 1, 1])) / det
        gy = (P[:, 0] * (x[:, 2, 0] - x[:, 1, 0]) + P[:, 1] * (x[:, 0, 0] - x[:, 2, 0])
              + P[:, 2] * (x[:, 1, 0] - x[:, 0, 0])) / det
        f = 1e21 * np.sqrt(gx * gx + gy * gy) / self.N0
        rhs = np.bincount(m.cells.ravel(), weights=np.repeat(f * np.abs(det) / 6.0, 3),
                          minlength=m.num_vertices())
        return self._mass.solve(rhs)

    def _install_device_pipeline(self):
        """The per-step refresh of fedm_gd.py:424-443 as field programs for the device."""
        ns, nr, N0 = self.ns, self.nr, self.N0
        tables, progs = [], []

        def table(x, y):
            tables.append((x, y))
            return len(tables) - 1

        arg = {"Umean": "energy", "E/N": "redfield"}
        for dep, kx, ky in zip(self.mu_dep, self.mu_x, self.mu_y):                  # mu rows
            progs.append(dict(kind="table", table=table(kx, ky), arg=arg[dep], scale=1.0 / N0)
                         if dep in arg else dict(kind="keep"))
        for i, (dep, kx, ky) in enumerate(zip(self.D_dep, self.D_x, self.D_y)):     # D rows
            if dep in arg:
                progs.append(dict(kind="table", table=table(kx, ky), arg=arg[dep], scale=1.0 / N0))
            elif dep == "ESR":
                progs.append(dict(kind="scaled_row", src_row=i, scale=kB * self.Tgas / elementary_charge))
            else:
                progs.append(dict(kind="keep"))
        for i in range(ns):                                                          # mu_diff rows
            progs.append(dict(kind="table", table=table(self.mu_x[i], self.mue_diff), arg="energy")
                         if i == ns - 1 else dict(kind="keep"))
        for i in range(ns):                                                          # D_diff rows
            progs.append(dict(kind="table", table=table(self.D_x[i], self.De_diff), arg="energy")
                         if i == ns - 1 else dict(kind="keep"))
        for dep, kx, ky in zip(self.k_dep, self.k_x, self.k_y):                      # k rows
            progs.append(dict(kind="table", table=table(kx, ky), arg=arg[dep]) if dep in arg
                         else dict(kind="keep"))
        for j, dep in enumerate(self.k_dep):                                         # k_diff rows
            progs.append(dict(kind="table", table=table(self.k_x[j], self.k_diff[j]), arg="energy")
                         if dep == "Umean" else dict(kind="keep"))
        progs += [dict(kind="me_old"), dict(kind="me"), dict(kind="ue_old")]
        self.prob.gd_prep_setup(tables, progs)

    def upload_fields(self):
        ns, nr, nv = self.ns, self.nr, self.mesh.num_vertices()
        me_old = self.mean_energy_old.vector()
        zeros = np.zeros(nv)
        mu_d = [zeros] * ns
        D_d = [zeros] * ns
        mu_d[ns - 1] = np.interp(me_old, self.mu_x[ns - 1], self.mue_diff)      # fedm_gd.py:442
        D_d[ns - 1] = np.interp(me_old, self.D_x[ns - 1], self.De_diff)
        k_d = [np.interp(me_old, self.k_x[j], self.k_diff[j]) if self.k_dep[j] == "Umean" else zeros
               for j in range(nr)]
        rows = [f.vector() for f in self.mu] + [f.vector() for f in self.D] + mu_d + D_d \
            + [f.vector() for f in self.k] + k_d + [me_old, self.mean_energy.vector(), self.U[:, ns - 1]]
        self.prob.set_gd_fields(np.stack(rows))

    def step(self):
        import contextlib, io, sys
        prob, ns = self.prob, self.ns
        t_old = self.t
        prob.shift_state()                                           # :422-423
        if self.device_pipeline:
            return self._step_on_device(t_old)
        U_old = self.U
        self.mean_energy_old.assign(self.mean_energy)
        self.redE.vector()[:] = self.project_reduced_field(self.U[:, ns])
        N0 = self.N0
        ff.Transport_coefficient_interpolation("update", self.mu_dep, N0, self.Tgas, self.mu, self.mu_x,
                                               self.mu_y, self.mean_energy_old, self.redE)
        ff.Transport_coefficient_interpolation("update", self.D_dep, N0, self.Tgas, self.D, self.D_x,
                                               self.D_y, self.mean_energy_old, self.redE, self.mu)
        ff.Rate_coefficient_interpolation("update", self.k_dep, self.k, self.k_x, self.k_y,
                                          self.mean_energy_old, self.redE, Te=0, Tgas=0)
        self.upload_fields()
        with contextlib.redirect_stdout(io.StringIO() if self.quiet else sys.stdout):
            self.t = ff.adaptive_solver(self.solver, self.problem, self.t, self.dt, self.dt_old,
                                        self.u_new, self.u_old, None, None, self.assigner, self.error,
                                        self.error_file, self.max_error, self.ttol, self.dt_min,
                                        time_dependent_arguments=[self.Phi_powered], approximation="LMEA")
        self.newton_iterations += prob.last_report.iterations
        self.linear_iterations += prob.last_report.linear_iterations
        self.U = prob.get_state()
        self.mean_energy.vector()[:] = np.exp(self.U[:, 0] - self.U[:, ns - 1])      # :452
        if self.snapshot is None and self.t_output <= self.t:        # file_output, file_io.py:582-587
            self.snapshot = U_old + (self.t_output - t_old) * (self.U - U_old) / (self.t - t_old)
        self.dt_old.time_step = self.dt.time_step
        self.dt.time_step = ff.adaptive_timestep(self.dt.time_step, self.max_error, self.ttol,
                                                 self.dt_min, self.dt_max)
        self.max_error[2] = self.max_error[1]
        self.max_error[1] = self.max_error[0]
        return self.t

    def _solve(self):
        import contextlib, io, sys
        with contextlib.redirect_stdout(io.StringIO() if self.quiet else sys.stdout):
            self.t = ff.adaptive_solver(self.solver, self.problem, self.t, self.dt, self.dt_old,
                                        self.u_new, self.u_old, None, None, self.assigner, self.error,
                                        self.error_file, self.max_error, self.ttol, self.dt_min,
                                        time_dependent_arguments=[self.Phi_powered], approximation="LMEA")
        self.newton_iterations += self.prob.last_report.iterations
        self.linear_iterations += self.prob.last_report.linear_iterations

    def _step_on_device(self, t_old):
        """Same step with the coefficient refresh and the mean-energy update on the device: no
        state leaves the GPU unless an output time is crossed."""
        prob = self.prob
        prob.gd_prep_step()
        self._solve()
        prob.gd_update_mean_energy()
        if self.snapshot is None and self.t_output <= self.t:
            U_old, U = prob.get_state_old(), prob.get_state()
            self.snapshot = U_old + (self.t_output - t_old) * (U - U_old) / (self.t - t_old)
        self.dt_old.time_step = self.dt.time_step
        self.dt.time_step = ff.adaptive_timestep(self.dt.time_step, self.max_error, self.ttol,
                                                 self.dt_min, self.dt_max)
        self.max_error[2] = self.max_error[1]
        self.max_error[1] = self.max_error[0]
        return self.t

    def run(self):
        while self.t < self.T_final:
            self.step()
        if self.device_pipeline:
            self.U = self.prob.get_state()
        return dict(log=[tuple(float(v) for v in line.split()) for line in open(self.error_file)],
                    snapshot=self.snapshot, U=self.U, t=self.t)
