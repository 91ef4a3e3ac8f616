"""Residual and Jacobian of FEDM's LFA model family on P1 triangles (oracle; test infra).

Restates, vectorised over cells with numpy, the integrals that the reference
builds symbolically and hands to ``dolfin.assemble``:

* balance equation in log variables  fedm/functions.py:350-368
  (variable-step BDF2 time term, flux term, source term, ``2*pi*r`` weight)
* drift-diffusion flux               fedm/functions.py:219-237
* Poisson equation                   fedm/functions.py:401
* Neumann boundary flux              fedm/functions.py:523-524
* Jacobian = exact Gateaux derivative (``derivative(F, u_new, u)``,
  examples/streamer_discharge/fedm-streamer.py:288-289), hand-derived here
* Dirichlet rows after assembly      fedm/functions.py:188-202

Unknowns are interleaved per vertex: dof = vertex * n_eq + component, species
first, potential last.  P1 makes grad(Phi) -- hence E, |E| and every
coefficient that depends on |E| -- constant per cell.
"""
import numpy as np
import scipy.sparse as sp

from .lagrange import interpolation_matrix, p1_basis
from .quadrature import interval_rule, triangle_rule

elementary_charge = 1.6021766208e-19   # fedm/physical_constants.py:5
epsilon_0 = 8.854187817e-12            # fedm/physical_constants.py:7

REACTION, DIFFUSION_REACTION, DRIFT_DIFFUSION_REACTION = 0, 1, 2
EQUATION_TYPES = {"reaction": REACTION, "diffusion-reaction": DIFFUSION_REACTION,
                  "drift-diffusion-reaction": DRIFT_DIFFUSION_REACTION}


class TermSum:
    """f(E) = sum_i c_i * E**p_i * exp(q_i * E**r_i)  and its derivative."""

    def __init__(self, terms):
        self.terms = [tuple(float(v) for v in t) for t in terms]   # (c, p, q, r)

    @classmethod
    def const(cls, c):
        return cls([(c, 0.0, 0.0, 0.0)])

    def __call__(self, E):
        E = np.asarray(E, dtype=np.float64)
        val = np.zeros_like(E)
        der = np.zeros_like(E)
        for c, p, q, r in self.terms:
            if c == 0.0:
                continue
            if p == 0.0 and q == 0.0:
                val = val + c
                continue
            lnE = np.log(E)
            g = q * np.exp(r * lnE) if q != 0.0 else 0.0          # q * E**r
            t = c * np.exp(p * lnE + g)
            val = val + t
            der = der + t * (p + r * g) / E
        return val, der


class LFAModel:
    """Species balance equations (+ optional Poisson row) of the LFA family."""

    def __init__(self, mesh, n_species, poisson, eq_type, Z, mu=None, D=None,
                 drift_w=None, reactions=(), facet_tags=None, bc_type=None,
                 qdeg=2, qdeg_time=None, qdeg_flux=None, qdeg_source=None,
                 qdeg_ext=None, axisymmetric=True, log_representation=True):
        self.mesh = mesh
        self.ns = n_species
        self.poisson = bool(poisson)
        self.neq = n_species + (1 if poisson else 0)
        self.eq_type = [EQUATION_TYPES[e] if isinstance(e, str) else int(e) for e in eq_type]
        self.Z = [float(z) for z in Z]
        as_ts = lambda v: v if isinstance(v, TermSum) else TermSum.const(v)
        self.mu = [as_ts(m) for m in (mu if mu is not None else [0.0] * n_species)]
        self.D = [as_ts(d) for d in (D if D is not None else [0.0] * n_species)]
        self.drift_w = list(drift_w) if drift_w is not None else [None] * n_species
        # reactions: (k TermSum|float, power[ns], net[ns])
        self.reactions = [(as_ts(k), list(P), list(nu)) for k, P, nu in reactions]
        self.facet_tags = facet_tags
        self.bc_type = bc_type            # bc_type[tag-1][s]
        self.qdeg_time = qdeg if qdeg_time is None else qdeg_time
        self.qdeg_flux = qdeg if qdeg_flux is None else qdeg_flux
        self.qdeg_source = qdeg if qdeg_source is None else qdeg_source
        self.qdeg_ext = qdeg if qdeg_ext is None else qdeg_ext
        self.qdeg_facet = qdeg
        self.ext_source = [None] * n_species     # (degree k, nodal values (Nc, nnodes))
        self.dirichlet_dofs = np.zeros(0, dtype=np.int64)
        self.dirichlet_vals = np.zeros(0, dtype=np.float64)
        self.axisymmetric = axisymmetric
        # fedm/functions.py:350-368: log_representation=False solves for the densities themselves
        # (time term u_part/dt without the exp(u) factor, flux -grad(D u) + sign mu E u, sources and
        # the Poisson charge in terms of u)
        self.log = bool(log_representation)
        self._geometry()
        self._pattern()

    # ------------------------------------------------------------------
    def _geometry(self):
        x = self.mesh.coords[self.mesh.cells]                     # (Nc,3,2)
        d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
        self.detJ = np.abs(det)                                   # = 2*area
        G = np.empty_like(x)
        # grad phi_a = rot90(edge opposite a) / det
        G[:, 0, 0] = (x[:, 1, 1] - x[:, 2, 1]) / det
        G[:, 0, 1] = (x[:, 2, 0] - x[:, 1, 0]) / det
        G[:, 1, 0] = (x[:, 2, 1] - x[:, 0, 1]) / det
        G[:, 1, 1] = (x[:, 0, 0] - x[:, 2, 0]) / det
        G[:, 2, 0] = (x[:, 0, 1] - x[:, 1, 1]) / det
        G[:, 2, 1] = (x[:, 1, 0] - x[:, 0, 0]) / det
        self.G = G
        if self.axisymmetric:
            self.rnod = x[:, :, 0].copy()                         # r = x[0]
        else:
            self.rnod = np.full(x.shape[:2], 0.5 / np.pi)         # functions.py:251
        self.xc = x

    def _pattern(self):
        c = self.mesh.cells.astype(np.int64)
        neq = self.neq
        rows = (c[:, :, None, None, None] * neq + np.arange(neq)[None, None, :, None, None])
        cols = (c[:, None, None, :, None] * neq + np.arange(neq)[None, None, None, None, :])
        shape = (c.shape[0], 3, neq, 3, neq)
        self._rows = np.broadcast_to(rows, shape).ravel()
        self._cols = np.broadcast_to(cols, shape).ravel()

    def set_dirichlet(self, dofs, vals):
        self.dirichlet_dofs = np.asarray(dofs, dtype=np.int64)
        self.dirichlet_vals = np.asarray(vals, dtype=np.float64)

    def set_ext_source(self, s, degree, nodal):
        self.ext_source[s] = (degree, np.asarray(nodal, dtype=np.float64))

    # ------------------------------------------------------------------
    def cell_fields(self, U):
        """Per-cell E, |E| from the potential column (zero field if no Poisson)."""
        Uc = U[self.mesh.cells]                                   # (Nc,3,neq)
        if self.poisson:
            gradPhi = np.einsum("ca,cad->cd", Uc[:, :, self.neq - 1], self.G)
            E = -gradPhi
            Em = np.sqrt(np.einsum("cd,cd->c", E, E))
        else:
            E = np.zeros((self.mesh.nc, 2))
            Em = np.ones(self.mesh.nc)
        return Uc, E, Em

    def _dens(self, uq):
        """density and d(density)/du at quadrature points: exp(u), exp(u) or u, 1"""
        if self.log:
            n = np.exp(uq)
            return n, n
        return uq, np.ones_like(uq)

    def element_tensors(self, U, Uold, Uold1, dt, dt_old, jacobian=True):
        nc, ns, neq, G = self.mesh.nc, self.ns, self.neq, self.G
        Uc, E, Em = self.cell_fields(U)
        Uoc, Uo1c = Uold[self.mesh.cells], Uold1[self.mesh.cells]
        Re = np.zeros((nc, 3, neq))
        Ke = np.zeros((nc, 3, neq, 3, neq)) if jacobian else None
        iphi = neq - 1
        two_pi = 2.0 * np.pi

        # time-step ratio factors, fedm/functions.py:350-356
        tr = dt / dt_old
        trp1 = 1.0 + tr
        tr2p1 = 1.0 + 2.0 * tr

        gradu = np.einsum("cas,cad->csd", Uc, G)                  # (Nc,neq,2)
        GG = np.einsum("cad,cbd->cab", G, G)
        if self.poisson:
            dEm = -np.einsum("cd,cbd->cb", E, G) / Em[:, None]    # dEm/dPhi_b
        mu = [m(Em) for m in self.mu]
        Dc = [d(Em) for d in self.D]
        kk = [k(Em) for k, _, _ in self.reactions]

        def point(xi, w):
            phi = p1_basis(xi[None, :])[0]
            rq = self.rnod @ phi
            return phi, w * self.detJ * two_pi * rq

        # ---- time-derivative term --------------------------------------
        xq, wq = triangle_rule(self.qdeg_time)
        for xi, w in zip(xq, wq):
            phi, W = point(xi, w)
            for s in range(ns):
                u = Uc[:, :, s] @ phi
                uo = Uoc[:, :, s] @ phi
                uo1 = Uo1c[:, :, s] @ phi
                u_part = (u * tr2p1 - trp1 ** 2.0 * uo + tr ** 2.0 * uo1) / trp1
                if self.log:
                    n = np.exp(u)
                    T = n * u_part / dt
                    dT = n * (u_part / dt + tr2p1 / (trp1 * dt))
                else:                                   # expu_or_1 = 1.0, functions.py:352
                    T = u_part / dt
                    dT = np.full_like(u, tr2p1 / (trp1 * dt))
                for a in range(3):
                    Re[:, a, s] += W * T * phi[a]
                    if jacobian:
                        for b in range(3):
                            Ke[:, a, s, b, s] += W * dT * phi[a] * phi[b]

        # ---- flux term -------------------------------------------------
        xq, wq = triangle_rule(self.qdeg_flux)
        for xi, w in zip(xq, wq):
            phi, W = point(xi, w)
            for s in range(ns):
                if self.eq_type[s] == REACTION:
                    continue
                n, dn = self._dens(Uc[:, :, s] @ phi)
                Dv, Dd = Dc[s]
                # Gamma = -D grad(u_e) + sign mu E u_e with u_e = exp(u) (grad u_e = u_e grad u) or u
                gn = gradu[:, s, :] * (n[:, None] if self.log else 1.0)      # grad(u_e)
                flux = -Dv[:, None] * gn
                field_drift = False
                drift = None
                if self.eq_type[s] == DRIFT_DIFFUSION_REACTION:
                    if self.drift_w[s] is not None:
                        drift = np.broadcast_to(np.asarray(self.drift_w[s], dtype=np.float64)[None, :], flux.shape)
                    elif self.poisson:
                        muv, mud = mu[s]
                        drift = (self.Z[s] * muv)[:, None] * E
                        field_drift = True
                if drift is not None:
                    flux = flux + drift * n[:, None]
                for a in range(3):
                    Re[:, a, s] -= W * np.einsum("cd,cd->c", flux, G[:, a])
                    if not jacobian:
                        continue
                    for b in range(3):
                        # d grad(u_e)/du_b = dn phi_b grad u + n G_b (log) or G_b (linear)
                        if self.log:
                            dgn = (dn * phi[b])[:, None] * gradu[:, s, :] + n[:, None] * G[:, b]
                        else:
                            dgn = G[:, b]
                        dflux = -Dv[:, None] * dgn
                        if drift is not None:
                            dflux = dflux + drift * (dn * phi[b])[:, None]
                        Ke[:, a, s, b, s] -= W * np.einsum("cd,cd->c", dflux, G[:, a])
                        if self.poisson:
                            dphi = -(Dd * dEm[:, b])[:, None] * gn
                            if field_drift:
                                dphi = dphi + ((self.Z[s] * mud * dEm[:, b])[:, None] * E
                                               - (self.Z[s] * muv)[:, None] * G[:, b]) * n[:, None]
                            Ke[:, a, s, b, iphi] -= W * np.einsum("cd,cd->c", dphi, G[:, a])

        # ---- reaction sources and Poisson ------------------------------
        xq, wq = triangle_rule(self.qdeg_source)
        for xi, w in zip(xq, wq):
            phi, W = point(xi, w)
            dens = [self._dens(Uc[:, :, s] @ phi) for s in range(ns)]
            nq = [d[0] for d in dens]
            dnq = [d[1] for d in dens]
            for (kv, kd), (_, P, nu) in zip(kk, self.reactions):
                prod = np.ones(nc)
                for i in range(ns):
                    if P[i]:
                        prod = prod * nq[i] ** P[i]
                # d prod / d u_i = P_i n_i^(P_i - 1) dn_i prod_(k != i) n_k^P_k
                dprod = []
                for i in range(ns):
                    if not P[i]:
                        dprod.append(None)
                        continue
                    d = P[i] * nq[i] ** (P[i] - 1) * dnq[i]
                    for k2 in range(ns):
                        if k2 != i and P[k2]:
                            d = d * nq[k2] ** P[k2]
                    dprod.append(d)
                for s in range(ns):
                    if nu[s] == 0:
                        continue
                    for a in range(3):
                        Re[:, a, s] -= W * nu[s] * kv * prod * phi[a]
                        if not jacobian:
                            continue
                        for b in range(3):
                            for i in range(ns):
                                if P[i]:
                                    Ke[:, a, s, b, i] -= W * nu[s] * kv * dprod[i] * phi[a] * phi[b]
                            if self.poisson:
                                Ke[:, a, s, b, iphi] -= W * nu[s] * kd * dEm[:, b] * prod * phi[a]
            if self.poisson:
                rho = np.zeros(nc)
                for s in range(ns):
                    rho = rho + self.Z[s] * nq[s] * elementary_charge / epsilon_0
                for a in range(3):
                    Re[:, a, iphi] += W * (np.einsum("cd,cd->c", -E, G[:, a]) - rho * phi[a])
                    if not jacobian:
                        continue
                    for b in range(3):
                        Ke[:, a, iphi, b, iphi] += W * GG[:, a, b]
                        for s in range(ns):
                            Ke[:, a, iphi, b, s] -= W * self.Z[s] * dnq[s] \
                                * elementary_charge / epsilon_0 * phi[a] * phi[b]

        # ---- interpolated (Expression) sources -------------------------
        for s in range(ns):
            if self.ext_source[s] is None:
                continue
            k, nodal = self.ext_source[s]
            xq, wq = triangle_rule(self.qdeg_ext)
            B = interpolation_matrix(k, xq)
            for q, (xi, w) in enumerate(zip(xq, wq)):
                phi, W = point(xi, w)
                f = nodal @ B[q]
                for a in range(3):
                    Re[:, a, s] -= W * f * phi[a]

        # ---- boundary flux terms (Neumann), functions.py:523-524 --------
        if self.facet_tags is not None and self.bc_type is not None and self.poisson:
            self._boundary(Uc, E, Em, dEm, mu, Re, Ke)
        return Re, Ke

    def _boundary(self, Uc, E, Em, dEm, mu, Re, Ke):
        ends = {0: (1, 2), 1: (0, 2), 2: (0, 1)}
        tq, wt = interval_rule(self.qdeg_facet)
        two_pi = 2.0 * np.pi
        for i in range(3):
            tagged = np.nonzero(self.facet_tags[:, i] > 0)[0]
            if tagged.size == 0:
                continue
            j, k = ends[i]
            for s in range(self.ns):
                kinds = np.array([self.bc_type[t - 1][s] == "Neumann"
                                  for t in self.facet_tags[tagged, i]])
                if self.eq_type[s] != DRIFT_DIFFUSION_REACTION or not kinds.any():
                    continue
                c = tagged[kinds]
                Gi = self.G[c, i]
                nrm = -Gi / np.linalg.norm(Gi, axis=1)[:, None]       # outward normal
                L = np.linalg.norm(self.xc[c, j] - self.xc[c, k], axis=1)
                muv, mud = mu[s][0][c], mu[s][1][c]
                En = np.einsum("cd,cd->c", E[c], nrm)
                for t, w in zip(tq, wt):
                    phi = np.zeros(3)
                    phi[j], phi[k] = 1.0 - t, t
                    rq = self.rnod[c] @ phi
                    n, dn = self._dens(Uc[c, :, s] @ phi)
                    W = w * L * two_pi * rq
                    for a in (j, k):
                        Re[c, a, s] += W * self.Z[s] * muv * En * n * phi[a]
                        if Ke is None:
                            continue
                        for b in range(3):
                            Ke[c, a, s, b, s] += W * self.Z[s] * muv * En * dn * phi[a] * phi[b]
                            dflux = mud * dEm[c, b] * En \
                                - muv * np.einsum("cd,cd->c", self.G[c, b], nrm)
                            Ke[c, a, s, b, self.neq - 1] += W * self.Z[s] * dflux * n * phi[a]

    # ------------------------------------------------------------------
    def residual(self, U, Uold, Uold1, dt, dt_old, apply_bc=True):
        Re, _ = self.element_tensors(U, Uold, Uold1, dt, dt_old, jacobian=False)
        F = self._scatter_vec(Re)
        if apply_bc and self.dirichlet_dofs.size:
            F[self.dirichlet_dofs] = U.ravel()[self.dirichlet_dofs] - self.dirichlet_vals
        return F

    def residual_jacobian(self, U, Uold, Uold1, dt, dt_old, apply_bc=True):
        Re, Ke = self.element_tensors(U, Uold, Uold1, dt, dt_old, jacobian=True)
        F = self._scatter_vec(Re)
        N = self.mesh.nv * self.neq
        J = sp.coo_matrix((Ke.ravel(), (self._rows, self._cols)), shape=(N, N)).tocsr()
        if apply_bc and self.dirichlet_dofs.size:
            F[self.dirichlet_dofs] = U.ravel()[self.dirichlet_dofs] - self.dirichlet_vals
            J = self._dirichlet_rows(J)
        return F, J

    def _scatter_vec(self, Re):
        N = self.mesh.nv * self.neq
        idx = (self.mesh.cells.astype(np.int64)[:, :, None] * self.neq
               + np.arange(self.neq)[None, None, :]).ravel()
        return np.bincount(idx, weights=Re.ravel(), minlength=N)

    def _dirichlet_rows(self, J):
        N = J.shape[0]
        keep = np.ones(N)
        keep[self.dirichlet_dofs] = 0.0
        J = sp.diags(keep) @ J
        ident = sp.coo_matrix((np.ones(self.dirichlet_dofs.size),
                               (self.dirichlet_dofs, self.dirichlet_dofs)), shape=(N, N))
        return (J + ident).tocsr()
