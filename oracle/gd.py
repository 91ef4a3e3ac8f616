"""Glow discharge in argon, LMEA, 4 particles / 5 equations (oracle; test infra).

Restates the reference's test harness
tests/integrated_tests/glow_discharge/fedm_gd.py (which is
examples/glow_discharge/fedm-gd.py with T_final = 1e-11 s) on numpy: nodal P1
coefficient fields from the deck tables (fedm/functions.py:531-750), their
semi-implicit linearisation in the mean energy (:753-774), reaction and energy
source terms (:777-912), drift-diffusion fluxes in log variables (:219-237),
'flux source' boundary conditions with reflection and secondary emission
(:514-522), Poisson (:401), time-dependent Dirichlet voltage, variable-step
BDF2 (:350-357) and the adaptive loop (:958-1130).

The residual is written once, on values-with-spatial-gradient objects (:class:`SG`);
the Jacobian is its complex-step derivative per local dof -- exact to rounding,
like UFL's symbolic ``derivative`` (fedm_gd.py:402), without a hand derivation.
``abs`` (Max/Min and the |mu E.n| wall flux) is made complex-safe by its real part.
"""
from pathlib import Path

import numpy as np
import scipy.sparse as sp

from . import controller
from .lagrange import p1_basis
from .mesh import mark_boundaries, rectangle_crossed
from .newton import direct_solve
from .quadrature import interval_rule, triangle_rule

elementary_charge = 1.6021766208e-19
epsilon_0 = 8.854187817e-12
kB = 1.38064852e-23
kB_eV = 8.6173303e-5


def cabs(x):
    """|x| that stays analytic for complex-step differentiation."""
    return np.where(np.real(x) < 0, -x, x)


class SG:
    """A scalar field at one quadrature point of every cell: value + (d/dr, d/dz)."""
    __array_priority__ = 100

    def __init__(self, v, gx=0.0, gy=0.0):
        self.v, self.gx, self.gy = v, gx, gy

    @staticmethod
    def lift(x):
        return x if isinstance(x, SG) else SG(x)

    def __add__(self, o):
        o = SG.lift(o)
        return SG(self.v + o.v, self.gx + o.gx, self.gy + o.gy)

    __radd__ = __add__

    def __neg__(self):
        return SG(-self.v, -self.gx, -self.gy)

    def __sub__(self, o):
        return self + (-SG.lift(o))

    def __rsub__(self, o):
        return SG.lift(o) - self

    def __mul__(self, o):
        o = SG.lift(o)
        return SG(self.v * o.v, self.gx * o.v + self.v * o.gx, self.gy * o.v + self.v * o.gy)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = SG.lift(o)
        inv = 1.0 / o.v
        return SG(self.v * inv, (self.gx - self.v * inv * o.gx) * inv, (self.gy - self.v * inv * o.gy) * inv)

    def __rtruediv__(self, o):
        return SG.lift(o) / self

    def exp(self):
        e = np.exp(self.v)
        return SG(e, e * self.gx, e * self.gy)


def energy_loss_factor(loss, Ei, mean_energy):
    """fedm/functions.py:905-911: the energy an electron loses in a reaction -- the deck's number, or for its two
    sentinel values (Ei - mean energy) and the mean energy itself."""
    if 7e77 < loss < 8e77:
        return Ei - mean_energy
    if 9e99 < loss < 1e100:
        return mean_energy
    return loss


class Deck:
    """The 4_particles deck through the reference's file formats (read with numpy only)."""

    def __init__(self, path):
        path = Path(path)
        self.species_files = ["Ar_1p0", "Ar_star", "Ar_plus", "electrons"]
        self.names = ["Ar[1p0]", "Ar[*]", "Ar[+]", "e"]
        tab = lambda f: np.loadtxt(f, comments="#")
        tc = path / "transport_coefficients"
        self.mu_ion = tab(tc / "Ar_plus_Nb.dat")          # E/N
        self.mu_e = tab(tc / "electrons_Nb.dat")          # Umean
        self.D_e = tab(tc / "electrons_ND.dat")           # Umean
        self.D_const = [float(tab(tc / "Ar_1p0_ND.dat")), float(tab(tc / "Ar_star_ND.dat"))]
        rc = path / "rate_coefficients"
        self.k_tab = [tab(rc / f"k_00{i}.dat") for i in range(1, 5)]
        self.k5 = float(tab(rc / "k_005.dat"))
        self.k6 = float(tab(rc / "k_ArStarLifetime.dat"))
        self.k_el = tab(rc / "Pelastic.dat")
        self.energy_loss = [11.55, 15.76, -11.55, 4.21, -7.34, 0.0, 1.0]      # reacscheme.cfg Uin
        # Energy_Source_term's Ei (fedm/functions.py:855) for the sentinel losses 7.77e77 / 9.99e99 (:906-909); the deck
        # has none -- tests put them in
        self.energy_Ei = 0.0
        self.power = np.array([[1, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1], [0, 1, 0, 1],
                               [0, 2, 0, 0], [0, 1, 0, 0], [1, 0, 0, 1]])
        loss = np.array([[1, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0],
                         [0, 2, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0]])
        gain = np.array([[0, 1, 0, 0], [0, 0, 1, 1], [1, 0, 0, 0], [0, 0, 1, 1],
                         [1, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]])
        self.net = gain - loss
        self.M = [6.63352032e-26, 6.63352032e-26, 6.63352032e-26, 9.10938215e-31]
        self.sign = [0.0, 0.0, 1.0, -1.0]


class GlowDischarge:
    def __init__(self, deck_dir, nx=100, ny=100, U_w=-250.0, p0=1.0, Tgas=300.0, mesh=None):
        self.deck = Deck(deck_dir)
        d = self.deck
        self.N0 = p0 * 3.21877e22
        self.Tgas, self.U_w = Tgas, U_w
        self.gap = self.wall = 0.01
        self.mesh = mesh if mesh is not None else rectangle_crossed(0.0, 0.0, self.wall, self.gap, nx, ny)
        m = self.mesh
        self.tags = mark_boundaries(m, [["line", 0.0, 0.0, 0.0, self.wall],
                                        ["line", self.gap, self.gap, 0.0, self.wall],
                                        ["line", 0.0, self.gap, 0.0, 0.0],
                                        ["line", 0.0, self.gap, self.wall, self.wall]])
        self.ref = [[0.3, 0.3, 5e-4, 0.3]] * 2 + [[1.0] * 4] * 2       # fedm_gd.py:150-152
        self.gamma = [0.06, 0.06, 0.0, 0.0]
        self.we_met = 5.0
        self.neq = 5
        x = m.coords[m.cells]
        d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
        self.detJ = np.abs(det)
        G = np.empty_like(x)
        G[:, 0, 0] = (x[:, 1, 1] - x[:, 2, 1]) / det
        G[:, 0, 1] = (x[:, 2, 0] - x[:, 1, 0]) / det
        G[:, 1, 0] = (x[:, 2, 1] - x[:, 0, 1]) / det
        G[:, 1, 1] = (x[:, 0, 0] - x[:, 2, 0]) / det
        G[:, 2, 0] = (x[:, 0, 1] - x[:, 1, 1]) / det
        G[:, 2, 1] = (x[:, 1, 0] - x[:, 0, 0]) / det
        self.G, self.xc = G, x
        z = m.coords[:, 1]
        self.powered = np.nonzero(np.abs(z) <= 3e-16)[0]
        self.grounded = np.nonzero(np.abs(z - self.gap) <= 3e-16)[0]
        self.dirichlet_dofs = np.concatenate([self.powered, self.grounded]) * 5 + 4
        self.xq, self.wq = triangle_rule(4)
        self.tq, self.wt = interval_rule(4)
        # tables of derivatives, fedm_gd.py:83-96
        self.De_diff = np.gradient(d.D_e[:, 1], d.D_e[:, 0]) / self.N0
        self.mue_diff = np.gradient(d.mu_e[:, 1], d.mu_e[:, 0]) / self.N0
        self.k_diff = [np.gradient(t[:, 1], t[:, 0]) for t in d.k_tab] + [None, None,
                                                                            np.gradient(d.k_el[:, 1], d.k_el[:, 0])]
        self.vth_heavy = [0.0, np.sqrt(8.0 * kB * Tgas / (np.pi * d.M[1])),
                          np.sqrt(8.0 * kB * Tgas / (np.pi * d.M[2]))]
        c = m.cells.astype(np.int64)
        rows = (c[:, :, None, None, None] * 5 + np.arange(5)[None, None, :, None, None])
        cols = (c[:, None, None, :, None] * 5 + np.arange(5)[None, None, None, None, :])
        shape = (m.nc, 3, 5, 3, 5)
        self._rows = np.broadcast_to(rows, shape).ravel()
        self._cols = np.broadcast_to(cols, shape).ravel()
        self._mass = None

    # -- nodal coefficient fields, fedm/functions.py:621-637, 724-748 -----------------------
    def coefficients(self, mean_energy, redE):
        d, N0 = self.deck, self.N0
        nv = self.mesh.nv
        itp = lambda x, t: np.interp(x, t[:, 0], t[:, 1])
        mu = [np.zeros(nv), np.zeros(nv), itp(redE, d.mu_ion) / N0, itp(mean_energy, d.mu_e) / N0]
        D = [np.full(nv, d.D_const[0] / N0), np.full(nv, d.D_const[1] / N0),
             kB * self.Tgas * mu[2] / elementary_charge, itp(mean_energy, d.D_e) / N0]
        k = [itp(mean_energy, t) for t in d.k_tab] + [np.full(nv, d.k5), np.full(nv, d.k6),
                                                      itp(mean_energy, d.k_el)]
        kd = [np.interp(mean_energy, t[:, 0], g) for t, g in zip(d.k_tab, self.k_diff[:4])] + \
             [np.zeros(nv), np.zeros(nv), np.interp(mean_energy, d.k_el[:, 0], self.k_diff[6])]
        mu_d = np.interp(mean_energy, d.mu_e[:, 0], self.mue_diff)
        D_d = np.interp(mean_energy, d.D_e[:, 0], self.De_diff)
        return dict(mu=mu, D=D, k=k, kd=kd, mu_e_diff=mu_d, D_e_diff=D_d)

    def project_cell_field(self, f_cell):
        """project(f) for a cell-wise constant f: consistent P1 mass solve (Cartesian dx)."""
        m = self.mesh
        if self._mass is None:
            ref = (np.ones((3, 3)) + np.eye(3)) / 24.0
            vals = self.detJ[:, None, None] * ref[None]
            c = m.cells.astype(np.int64)
            r_ = np.broadcast_to(c[:, :, None], vals.shape).ravel()
            c_ = np.broadcast_to(c[:, None, :], vals.shape).ravel()
            import scipy.sparse.linalg as spla
            self._mass = spla.splu(sp.coo_matrix((vals.ravel(), (r_, c_)), shape=(m.nv, m.nv)).tocsc())
        rhs = np.bincount(m.cells.ravel(), weights=np.repeat(f_cell * self.detJ / 6.0, 3), minlength=m.nv)
        return self._mass.solve(rhs)

    def reduced_field(self, Phi):
        gp = np.einsum("ca,cad->cd", Phi[self.mesh.cells], self.G)
        return self.project_cell_field(1e21 * np.sqrt(np.einsum("cd,cd->c", gp, gp)) / self.N0)

    # -- element residual -------------------------------------------------------------------
    def element_residual(self, Uc, Uoc, Uo1c, dt, dt_old, co, me_old, me, ue_old):
        """Uc (Nc,3,5) may be complex.  co: nodal coefficient dict; me_old / me: nodal mean
        energies (old, current Function); ue_old: nodal old electron log density."""
        d, N0, G, cells = self.deck, self.N0, self.G, self.mesh.cells
        cdt = Uc.dtype
        R = np.zeros(Uc.shape, dtype=cdt)
        tr = dt / dt_old
        trp1, tr2p1 = 1.0 + tr, 1.0 + 2.0 * tr
        two_pi = 2.0 * np.pi
        nod = lambda a: a[cells]                                   # nodal -> (Nc,3)
        gradn = lambda A: (np.einsum("ca,ca->c", A, G[:, :, 0]), np.einsum("ca,ca->c", A, G[:, :, 1]))

        def fields(phi):
            """everything the integrands need at the point with P1 values phi, as SG"""
            def sg(A):
                gx, gy = gradn(A)
                return SG(A @ phi, gx, gy)
            u = [sg(Uc[:, :, i]) for i in range(5)]
            uo = [Uoc[:, :, i] @ phi for i in range(5)]
            uo1 = [Uo1c[:, :, i] @ phi for i in range(5)]
            meo, mec, ueo = sg(nod(me_old)), sg(nod(me)), sg(nod(ue_old))
            E = (-u[4].gx, -u[4].gy)
            n = [None] + [u[i].exp() for i in (1, 2, 3)]
            me_e = meo + (u[0].exp() - n[3] * meo) / ueo.exp()      # fedm_gd.py:215
            dme = me_e - meo
            mu = [sg(nod(co["mu"][i])) for i in range(4)]
            D = [sg(nod(co["D"][i])) for i in range(4)]
            mu[3] = mu[3] + sg(nod(co["mu_e_diff"])) * dme            # semi_implicit_coefficients
            D[3] = D[3] + sg(nod(co["D_e_diff"])) * dme
            k = []
            for j in range(7):
                kj = sg(nod(co["k"][j]))
                if j in (0, 1, 2, 3, 6):
                    kj = kj + sg(nod(co["kd"][j])) * dme
                k.append(kj)
            return u, uo, uo1, E, n, mu, D, k, mec

        def flux(sign, ulog, D, mu, E, grad_diffusion):
            """functions.py:219-237 -> (Gx, Gy) values"""
            ue = ulog.exp()
            if grad_diffusion:
                De = D * ue
                dx_, dy_ = -De.gx, -De.gy
            else:
                dx_, dy_ = -D.v * ue.gx, -D.v * ue.gy
            return dx_ + sign * mu.v * E[0] * ue.v, dy_ + sign * mu.v * E[1] * ue.v

        def bdf(u, uo, uo1):
            return (u.v * tr2p1 - trp1 ** 2.0 * uo + tr ** 2.0 * uo1) / trp1

        rn = self.xc[:, :, 0]
        for xi, w in zip(self.xq, self.wq):
            phi = p1_basis(xi[None, :])[0]
            W = w * self.detJ * two_pi * (rn @ phi)
            u, uo, uo1, E, n, mu, D, k, _ = fields(phi)
            expN = [N0, n[1].v, n[2].v, n[3].v]
            rate = []
            for j in range(7):
                t = k[j].v
                for i in range(4):
                    for _ in range(d.power[j, i]):
                        t = t * expN[i]
                rate.append(t)
            f = [sum(rate[j] * d.net[j, i] for j in range(7)) for i in range(4)]
            # fedm/functions.py:905-911 with the mean_energy argument of the scripts, u[0] / u[n - 1] (fedm-gd.py:358)
            f_en = sum(-rate[j] * energy_loss_factor(d.energy_loss[j], d.energy_Ei, u[0].v / u[3].v) for j in range(7))
            Ge = flux(d.sign[3], u[3], D[3], mu[3], E, True)
            f_en = f_en - (Ge[0] * E[0] + Ge[1] * E[1])                  # Joule heating, :359
            Gam = {1: None, 2: flux(d.sign[2], u[2], D[2], mu[2], E, False), 3: Ge}
            # Ar*: 'diffusion-reaction': -grad(D exp(u)), functions.py:362-364
            Ds = D[1] * u[1].exp()
            Gam[1] = (-Ds.gx, -Ds.gy)
            Gen = flux(d.sign[3], u[0], D[3] * (5.0 / 3.0), mu[3] * (5.0 / 3.0), E, True)
            rho = elementary_charge * (d.sign[2] * n[2].v + d.sign[3] * n[3].v) / epsilon_0
            for a in range(3):
                Ga0, Ga1 = G[:, a, 0], G[:, a, 1]
                for comp, (ulog, fl, src) in {1: (u[1], Gam[1], f[1]), 2: (u[2], Gam[2], f[2]),
                                               3: (u[3], Gam[3], f[3]), 0: (u[0], Gen, f_en)}.items():
                    T = np.exp(ulog.v) * bdf(ulog, uo[comp], uo1[comp]) / dt
                    R[:, a, comp] += W * (T * phi[a] - (fl[0] * Ga0 + fl[1] * Ga1) - src * phi[a])
                R[:, a, 4] += W * ((u[4].gx * Ga0 + u[4].gy * Ga1) - rho * phi[a])

        # ---- 'flux source' boundaries, functions.py:514-522 -------------------------------
        ends = {0: (1, 2), 1: (0, 2), 2: (0, 1)}
        for i in range(3):
            cs = np.nonzero(self.tags[:, i] > 0)[0]
            if cs.size == 0:
                continue
            j, k_ = ends[i]
            sub = _Sub(self, cs, Uc, Uoc, Uo1c)
            tagv = self.tags[cs, i]
            ref = np.array([self.ref[t - 1] for t in tagv])               # (nf,4)
            gam = np.array([self.gamma[t - 1] for t in tagv])
            Gi = G[cs, i]
            nrm = -Gi / np.linalg.norm(Gi, axis=1)[:, None]
            L = np.linalg.norm(self.xc[cs, j] - self.xc[cs, k_], axis=1)
            for t, w in zip(self.tq, self.wt):
                phi = np.zeros(3)
                phi[j], phi[k_] = 1.0 - t, t
                W = w * L * two_pi * (rn[cs] @ phi)
                u, E, n, mu, D, mec = sub.fields(phi, co, me_old, me, ue_old)
                En = E[0] * nrm[:, 0] + E[1] * nrm[:, 1]
                Gion = sub.flux(d.sign[2], u[2], D[2], mu[2], E, False)
                Ion = (Gion[0] * nrm[:, 0] + Gion[1] * nrm[:, 1])
                Ion = (Ion + 0.0 + cabs(Ion - 0.0)) / 2.0                # Max(., 0), functions.py:205-209
                vth_e = np.sqrt(16.0 * elementary_charge * mec.v / (3.0 * np.pi * d.M[3]))
                terms = {}
                fac = lambda s: (1.0 - ref[:, s]) / (1.0 + ref[:, s])
                terms[1] = fac(1) * 0.5 * self.vth_heavy[1] * n[1].v                       # diffusion-reaction
                terms[2] = fac(2) * (0.5 * self.vth_heavy[2] + cabs(d.sign[2] * mu[2].v * En)) * n[2].v
                terms[3] = fac(3) * (0.5 * vth_e + cabs(d.sign[3] * mu[3].v * En)) * n[3].v \
                    - 2.0 * gam * Ion / (1.0 + ref[:, 3])
                terms[0] = fac(3) * (0.5 * 1.3333 * vth_e + cabs(d.sign[3] * (5.0 / 3.0) * mu[3].v * En)) \
                    * np.exp(u[0].v) - 2.0 * (gam * self.we_met) * Ion / (1.0 + ref[:, 3])
                for a in (j, k_):
                    for comp, val in terms.items():
                        R[cs, a, comp] += W * val * phi[a]
        return R

    # -- global residual / Jacobian ------------------------------------------------------------
    def _scatter(self, Re):
        idx = (self.mesh.cells.astype(np.int64)[:, :, None] * 5 + np.arange(5)[None, None, :]).ravel()
        return np.bincount(idx, weights=Re.ravel(), minlength=self.mesh.nv * 5)

    def residual_jacobian(self, U, Uo, Uo1, dt, dt_old, co, me_old, me, ue_old, dir_vals, jacobian=True):
        cells = self.mesh.cells
        Uc, Uoc, Uo1c = U[cells], Uo[cells], Uo1[cells]
        Re = self.element_residual(Uc, Uoc, Uo1c, dt, dt_old, co, me_old, me, ue_old)
        F = self._scatter(Re)
        F[self.dirichlet_dofs] = U.ravel()[self.dirichlet_dofs] - dir_vals
        if not jacobian:
            return F, None
        h = 1e-30
        Ke = np.empty((self.mesh.nc, 3, 5, 3, 5))

        def column(bs):
            b, s = bs
            Up = Uc.astype(np.complex128)
            Up[:, b, s] += 1j * h
            Ke[:, :, :, b, s] = np.imag(self.element_residual(Up, Uoc, Uo1c, dt, dt_old, co,
                                                              me_old, me, ue_old)) / h

        from concurrent.futures import ThreadPoolExecutor
        import os
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            list(pool.map(column, [(b, s) for b in range(3) for s in range(5)]))
        N = self.mesh.nv * 5
        J = sp.coo_matrix((Ke.ravel(), (self._rows, self._cols)), shape=(N, N)).tocsr()
        keep = np.ones(N)
        keep[self.dirichlet_dofs] = 0.0
        J = sp.diags(keep) @ J + sp.coo_matrix((np.ones(self.dirichlet_dofs.size),
                                                (self.dirichlet_dofs, self.dirichlet_dofs)), shape=(N, N))
        return F, J.tocsr()

    def dirichlet_values(self, t):
        return np.concatenate([np.full(self.powered.size, self.U_w * (1.0 - np.exp(-t / 1e-9))),
                               np.zeros(self.grounded.size)])


class _Sub:
    """Point evaluation restricted to a subset of cells (boundary facets)."""

    def __init__(self, gd, cs, Uc, Uoc, Uo1c):
        self.gd, self.cs = gd, cs
        self.Uc = Uc[cs]
        self.G = gd.G[cs]
        self.cells = gd.mesh.cells[cs]

    def sg(self, A, phi):
        return SG(A @ phi, np.einsum("ca,ca->c", A, self.G[:, :, 0]), np.einsum("ca,ca->c", A, self.G[:, :, 1]))

    def fields(self, phi, co, me_old, me, ue_old):
        nod = lambda a: a[self.cells]
        sg = lambda A: self.sg(A, phi)
        u = [sg(self.Uc[:, :, i]) for i in range(5)]
        meo, mec, ueo = sg(nod(me_old)), sg(nod(me)), sg(nod(ue_old))
        E = (-u[4].gx, -u[4].gy)
        n = [None] + [u[i].exp() for i in (1, 2, 3)]
        dme = (meo + (u[0].exp() - n[3] * meo) / ueo.exp()) - meo
        mu = [sg(nod(co["mu"][i])) for i in range(4)]
        D = [sg(nod(co["D"][i])) for i in range(4)]
        mu[3] = mu[3] + sg(nod(co["mu_e_diff"])) * dme
        D[3] = D[3] + sg(nod(co["D_e_diff"])) * dme
        return u, E, n, mu, D, mec

    @staticmethod
    def flux(sign, ulog, D, mu, E, grad_diffusion):
        ue = ulog.exp()
        if grad_diffusion:
            De = D * ue
            dx_, dy_ = -De.gx, -De.gy
        else:
            dx_, dy_ = -D.v * ue.gx, -D.v * ue.gy
        return dx_ + sign * mu.v * E[0] * ue.v, dy_ + sign * mu.v * E[1] * ue.v


def run(deck_dir, T_final=1e-11, dt_init=1e-13, dt_max=1e-8, dt_min=1e-15, ttol=2e-3, rtol=1e-4,
        max_it=20, nx=100, ny=100, t_output=1e-11, verbose=False):
    """Time loop of fedm_gd.py:420-471.  Returns dict(log, snapshot (vertex order), model)."""
    gd = GlowDischarge(deck_dir, nx, ny)
    nv = gd.mesh.nv
    n_ic = [gd.N0, 1e12, 1e12, 1e12]
    me = np.full(nv, 3.0)
    U = np.zeros((nv, 5))
    U[:, 0] = np.log(3.0) + np.log(n_ic[3])
    for i in (1, 2, 3):
        U[:, i] = np.log(n_ic[i])
    # initial potential: equal ion/electron densities, zero applied voltage at t=0 -> Phi = 0
    Uo, Uo1 = U.copy(), np.zeros_like(U)
    me_old = me.copy()
    st = controller.StepState(dt_init, 1e30, n_error=5)
    t, out = 0.0, {}

    def newton(Uw, dt, dt_old, co, t_new):
        dv = gd.dirichlet_values(t_new)
        fn0, its = None, 0
        while True:
            F, _ = gd.residual_jacobian(Uw, Uo, Uo1, dt, dt_old, co, me_old, me, Uo[:, 3], dv,
                                        jacobian=False)
            fn = float(np.linalg.norm(F))
            if not np.isfinite(fn):
                raise RuntimeError("NaN in residual")
            if its == 0:
                fn0 = fn
                if fn < 1e-10:
                    return its
            elif fn < 1e-10 or fn <= rtol * fn0:
                return its
            if its >= max_it:
                raise RuntimeError("Newton did not converge")
            _, J = gd.residual_jacobian(Uw, Uo, Uo1, dt, dt_old, co, me_old, me, Uo[:, 3], dv)
            Uw += direct_solve(J, -F).reshape(Uw.shape)
            its += 1

    while t < T_final:
        t_old = t
        Uo1[:] = Uo
        Uo[:] = U
        me_old = me.copy()
        redE = gd.reduced_field(U[:, 4])
        co = gd.coefficients(me_old, redE)

        def solve(Uw, dt, dt_old):
            its = newton(Uw, dt, dt_old, co, t + dt)
            if verbose:
                print("t", t + dt, "dt", dt, "newton", its, flush=True)

        t = controller.adaptive_solve(solve, U, Uo, t, st, ttol, dt_min, error_component=0)
        me = np.exp(U[:, 0] - U[:, 3])
        # file_output: linear interpolation to the output time, file_io.py:582-587
        if t_output <= t and "snapshot" not in out:
            out["snapshot"] = Uo + (t_output - t_old) * (U - Uo) / (t - t_old)
            out["snapshot_t"] = t_output
        st.dt_old = st.dt
        st.dt = controller.adaptive_timestep(st.dt, st.max_error, ttol, dt_min, dt_max)
        st.max_error[2] = st.max_error[1]
        st.max_error[1] = st.max_error[0]
    out.update(log=st.log, U=U, model=gd, t=t)
    return out
