"""Lowering of an LMEA script's weak form onto the device model of the glow-discharge family.

``fedm_amd.functions`` keeps FEDM's call sequence (examples/glow_discharge/fedm-gd.py:196-402):
``semi_implicit_coefficients`` -> ``Flux`` -> ``Source_term`` / ``Energy_Source_term`` ->
``weak_form_balance_equation_log_representation`` -> ``Boundary_flux('flux source', ...)`` ->
``Problem``.  The pieces those calls return are descriptors; this module checks that together they
are the model the device kernels implement (csrc/gd.hip: energy equation on component 0 with 5/3 of
the electron coefficients and Joule heating, particle balances with nodal semi-implicit
coefficients, 'flux source' walls with reflection and ion-induced secondary emission, Poisson) and
extracts its parameters into :class:`fedm_amd.device.GdModel`.  Anything else is refused with a
message that says which part of the script has no device counterpart -- never silently replaced.

It also binds the script's host Functions (mobilities, diffusion coefficients, rate coefficients,
their derivative tables, mean energies) to the rows of the device's nodal field table, so that what
``Transport_coefficient_interpolation`` & co. write on the host is what the next solve reads.
"""
import numpy as np

from . import forms
from .forms import Function, FunctionComponent, Sym, Unknown, evaluate


class Unsupported(NotImplementedError):
    pass


def _no(what):
    raise Unsupported(f"fedm_amd (LMEA on the device): {what}")


# ---------------------------------------------------------------------------------------------
# descriptors returned by the facade functions
# ---------------------------------------------------------------------------------------------
class SemiImplicit:
    """c + c' (mean_energy_new - mean_energy_old) for 'Umean' entries, c otherwise
    (fedm/functions.py:753-774); ``scale`` collects constant factors (5/3 of the electron
    coefficients in the energy flux, fedm-gd.py:354)."""

    def __init__(self, c, dc, dependent, me_new, me_old, scale=1.0):
        self.c, self.dc, self.dependent = c, dc, dependent
        self.me_new, self.me_old, self.scale = me_new, me_old, scale

    def _scaled(self, f):
        return SemiImplicit(self.c, self.dc, self.dependent, self.me_new, self.me_old, self.scale * f)

    def __mul__(self, o):
        return self._scaled(float(o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._scaled(1.0 / float(o))


def coefficient(x):
    """(nodal Function, derivative Function or None, constant factor) of a coefficient argument."""
    if isinstance(x, SemiImplicit):
        return x.c, (x.dc if x.dependent else None), x.scale
    if isinstance(x, Function):
        return x, None, 1.0
    if isinstance(x, Sym) and x.op in ("mul", "div"):          # 5.0 * mu / 3.0 of a plain Function
        a, b = x.args
        if x.op == "mul" and isinstance(a, (int, float)):
            f, d, s = coefficient(b)
            return f, d, s * float(a)
        if isinstance(b, (int, float)):
            f, d, s = coefficient(a)
            return f, d, s * float(b) if x.op == "mul" else s / float(b)
    _no(f"a transport coefficient must be a nodal Function (optionally semi-implicit and scaled), got {type(x).__name__}")


class LmeaSource:
    """f_i = sum_j (G - L)_ji k_j prod n^P (fedm/functions.py:835-843) with nodal rate coefficients."""

    def __init__(self, column, p_matrix, l_matrix, g_matrix, k_coeffs, N0):
        self.column, self.k, self.N0 = column, list(k_coeffs), float(N0)
        self.P, self.L, self.G = (np.asarray(m, dtype=int) for m in (p_matrix, l_matrix, g_matrix))


class LmeaEnergySource:
    """-sum_j rate_j loss_j (fedm/functions.py:901-912) [+ Joule heating, fedm-gd.py:359]."""

    def __init__(self, p_matrix, k_coeffs, u_loss, N0, joule=None, mean_energy=None, Ei=0):
        self.P, self.k, self.loss, self.N0, self.joule = np.asarray(p_matrix, dtype=int), list(k_coeffs), list(u_loss), float(N0), joule
        self.mean_energy, self.Ei = mean_energy, Ei      # used by the sentinel losses only (functions.py:906-909)

    def __add__(self, o):
        # f_en += -dot(Flux(electrons), E)
        if isinstance(o, Sym) and o.op == "neg" and isinstance(o.args[0], Sym) and o.args[0].op == "flux_dot_field":
            if self.joule is not None:
                _no("more than one Joule-heating term in the energy source")
            return LmeaEnergySource(self.P, self.k, self.loss, self.N0, joule=o.args[0],
                                    mean_energy=self.mean_energy, Ei=self.Ei)
        if isinstance(o, (int, float)) and o == 0:
            return self
        _no("the energy source may be extended by  -dot(Flux(electrons), E)  only")

    __radd__ = __add__


class IonFlux:
    """sum over ions of Max(dot(Gamma_i, n), 0) (fedm-gd.py:347-352)."""

    def __init__(self, fluxes=()):
        self.fluxes = list(fluxes)

    def __add__(self, o):
        if isinstance(o, IonFlux):
            return IonFlux(self.fluxes + o.fluxes)
        if isinstance(o, (int, float)) and o == 0:
            return self
        return NotImplemented

    __radd__ = __add__


def positive_part(a, b):
    """Max(a, b) of the facade when an operand is symbolic: only Max(dot(Gamma, n), 0)."""
    if isinstance(a, Sym) and a.op == "normal_flux" and isinstance(b, (int, float)) and b == 0:
        return IonFlux([a.args[0]])
    _no("Max()/Min() of symbolic operands other than Max(dot(Gamma, normal), 0)")


# ---------------------------------------------------------------------------------------------
# compilation
# ---------------------------------------------------------------------------------------------
def is_lmea(pieces):
    return any(isinstance(getattr(p, "f", None), (LmeaSource, LmeaEnergySource)) for p in pieces)


def _close(a, b, what, rtol=1e-3):
    if abs(a - b) > rtol * max(abs(a), abs(b), 1e-300):
        _no(f"{what}: the script uses {a}, the device kernel {b}")


def _mean_energy_parts(me_new):
    """mean_energy_e = me_old + (exp(u[0]) - exp(u[e]) * me_old) / exp(u_old_e)   (fedm-gd.py:215)"""
    try:
        assert me_new.op == "add"
        me_old, frac = me_new.args
        assert isinstance(me_old, Function) and frac.op == "div"
        num, den = frac.args
        assert den.op == "exp" and isinstance(den.args[0], Function)
        assert num.op == "sub" and isinstance(num.args[0], forms.Density) and num.args[0].index == 0
        prod = num.args[1]
        assert prod.op == "mul" and isinstance(prod.args[0], forms.Density) and prod.args[1] is me_old
        return me_old, den.args[0], prod.args[0].index
    except (AssertionError, AttributeError, IndexError, TypeError):
        _no("the semi-implicit mean energy must be  mean_energy_old + (exp(u[0]) - exp(u[e])*mean_energy_old)"
            "/exp(u_old_e)  (fedm-gd.py:215)")


class Binding:
    """Host Functions behind the rows of the device's nodal field table
    (mu, D, mu', D' per species; k, k' per reaction; mean_energy_old, mean_energy, u_old of the
    electrons): what `DeviceProblem.set_gd_fields` uploads before a solve."""

    def __init__(self, rows):
        self.rows = rows

    def stack(self, nv):
        zero = np.zeros(nv)
        return np.stack([zero if f is None else np.asarray(f.vector(), dtype=float) + zero for f in self.rows])


def compile_lmea(pieces, quadrature_degree):
    """Form pieces -> (GdModel, mesh, facet tags, Binding)."""
    from .device import GdModel
    from .functions import BalanceEq, BoundaryTerm, FluxDesc, PoissonEq
    from .physical_constants import elementary_charge as q_e, epsilon_0 as eps0
    balances = sorted([p for p in pieces if isinstance(p, BalanceEq)], key=lambda p: p.u.index)
    poissons = [p for p in pieces if isinstance(p, PoissonEq)]
    bterms = [p for p in pieces if isinstance(p, BoundaryTerm)]
    if len(poissons) != 1:
        _no("exactly one Poisson equation is needed")
    energy = [p for p in balances if isinstance(p.f, LmeaEnergySource)]
    if len(energy) != 1 or energy[0].u.index != 0:
        _no("the electron energy balance must be component 0 of the mixed space")
    energy = energy[0]
    parts = [p for p in balances if p is not energy]
    ns = len(parts) + 1                      # particle species incl. the background gas (species 0)
    if [p.u.index for p in parts] != list(range(1, ns)) or poissons[0].u.index != ns:
        _no("components must be [energy, species 1..n-1, potential] (fedm-gd.py:386-391)")
    if any(not isinstance(p.f, LmeaSource) for p in parts):
        _no("particle sources must come from Source_term(..., 'LMEA', ...) with nodal rate coefficients")
    space = energy.u.space
    mesh = space.mesh
    nv = mesh.num_vertices()
    ie = ns - 1                              # electrons are the last species
    src0 = parts[0].f
    nr = src0.P.shape[0]
    if src0.P.shape[1] != ns:
        _no("reaction matrices must have one column per species including the background gas")
    # ---- coefficients: one nodal Function (+ derivative) per species ------------------------
    mu_rows, D_rows, mud_rows, Dd_rows = [None] * ns, [None] * ns, [None] * ns, [None] * ns
    eq_type, sign, grad_diff = ["reaction"] * ns, [0.0] * ns, [False] * ns
    me_new = me_old = ue_old = None

    def note_semi(x):
        nonlocal me_new, me_old, ue_old
        if isinstance(x, SemiImplicit) and x.dependent:
            old, uo, e_idx = _mean_energy_parts(x.me_new)
            if x.me_old is not old or e_idx != ie:
                _no("semi-implicit coefficients must be linearised about mean_energy_old with the electron density last")
            if me_old is not None and (me_old is not old or ue_old is not uo):
                _no("all semi-implicit coefficients must share one mean-energy expression")
            me_new, me_old, ue_old = x.me_new, old, uo

    for p in parts:
        i = p.u.index
        eq_type[i] = p.equation_type
        if p.equation_type == "reaction":
            continue
        if p.equation_type == "diffusion-reaction":
            f, d, s = coefficient(p.D)
            _close(s, 1.0, f"scaling of D[{i}]")
            D_rows[i], Dd_rows[i] = f, d
            note_semi(p.D)
            if isinstance(p.Gamma, FluxDesc):
                fm, dm, _ = coefficient(p.Gamma.mu)
                mu_rows[i], mud_rows[i] = fm, dm
            continue
        g = p.Gamma
        if not isinstance(g, FluxDesc) or g.u is not p.u or not g.log:
            _no(f"species {i}: drift-diffusion-reaction needs Gamma = Flux(sign, u[{i}], D, mu, E, ..., logarithm_representation=True)")
        fm, dm, sm = coefficient(g.mu)
        fd, dd, sd = coefficient(g.D)
        _close(sm, 1.0, f"scaling of mu[{i}]")
        _close(sd, 1.0, f"scaling of D[{i}]")
        mu_rows[i], mud_rows[i], D_rows[i], Dd_rows[i] = fm, dm, fd, dd
        sign[i], grad_diff[i] = float(g.sign), bool(g.grad_diffusion)
        note_semi(g.mu)
        note_semi(g.D)
    if eq_type[ie] != "drift-diffusion-reaction":
        _no("the electrons (last species) must obey a drift-diffusion-reaction equation")
    # energy flux: 5/3 of the electron coefficients on u[0]
    ge = energy.Gamma
    if not isinstance(ge, FluxDesc) or ge.u is not energy.u:
        _no("the energy balance needs Gamma_en = Flux(sign_e, u[0], 5/3 D_e, 5/3 mu_e, E, ...)")
    fm, _, sm = coefficient(ge.mu)
    fd, _, sd = coefficient(ge.D)
    if fm is not mu_rows[ie] or fd is not D_rows[ie]:
        _no("the energy flux must use the electron mobility and diffusion coefficient")
    _close(sm, 5.0 / 3.0, "factor of the electron mobility in the energy flux")
    _close(sd, 5.0 / 3.0, "factor of the electron diffusion coefficient in the energy flux")
    _close(float(ge.sign), sign[ie], "sign of the energy flux")
    if energy.equation_type != "drift-diffusion-reaction" or bool(ge.grad_diffusion) != grad_diff[ie]:
        _no("the energy balance must be of the electrons' equation type and diffusion form")
    # Joule heating: -dot(Flux(electrons), E)
    j = energy.f.joule
    if j is None:
        _no("the energy source lacks the power input from the field, -dot(Flux(electrons), E) (fedm-gd.py:359)")
    jf = j.args[0]
    if jf.u.index != ie or coefficient(jf.mu)[0] is not mu_rows[ie] or coefficient(jf.D)[0] is not D_rows[ie]:
        _no("the Joule-heating term must be built from the electron flux")
    # ---- reactions ---------------------------------------------------------------------------
    k_rows, kd_rows = [None] * nr, [None] * nr
    ks = src0.k
    if len(ks) != nr or any(s.f.k is not ks and list(s.f.k) != list(ks) for s in parts) or list(energy.f.k) != list(ks):
        _no("all source terms must use the same list of rate coefficients")
    for jx, k in enumerate(ks):
        f, d, s = coefficient(k)
        _close(s, 1.0, f"scaling of rate coefficient {jx}")
        k_rows[jx], kd_rows[jx] = f, d
        note_semi(k)
    for s in parts:
        if s.f.column != s.u.index or not (np.array_equal(s.f.P, src0.P) and np.array_equal(s.f.G - s.f.L, src0.G - src0.L)):
            _no("source term f[i] must be used in the balance of species i with one set of reaction matrices")
    if abs(src0.N0 - energy.f.N0) > 0 or not np.array_equal(energy.f.P, src0.P):
        _no("particle and energy sources must share N0 and the power matrix")
    # Energy losses; the decks' sentinel values (fedm/functions.py:906-909): 7.77e77 -> Ei - mean_energy, 9.99e99 ->
    # mean_energy.  A numeric mean_energy is folded into the loss here; the expression the reference's scripts pass,
    # u[0] / u[n - 1] (fedm-gd.py:358), is evaluated and differentiated on the device; anything else is refused.
    loss = [float(v) for v in energy.f.loss]
    energy_Ei, me_form = 0.0, None
    if any(7e77 < v < 8e77 or 9e99 < v < 1e100 for v in loss):
        me_arg, Ei = energy.f.mean_energy, energy.f.Ei
        if isinstance(Ei, forms.Constant):
            Ei = Ei.value
        if not isinstance(Ei, (int, float)):
            _no("Energy_Source_term: Ei must be a number")
        energy_Ei = float(Ei)
        if isinstance(me_arg, forms.Constant):
            me_arg = me_arg.value
        if isinstance(me_arg, (int, float)):
            loss = [energy_Ei - float(me_arg) if 7e77 < v < 8e77 else float(me_arg) if 9e99 < v < 1e100 else v
                    for v in loss]
        elif (isinstance(me_arg, Sym) and me_arg.op == "div" and all(isinstance(a, forms.Unknown) for a in me_arg.args)
              and me_arg.args[0].index == 0 and me_arg.args[1].index == ns - 1):
            me_form = "unknown_ratio"
        else:
            _no("Energy_Source_term: with mean-energy-dependent losses, mean_energy must be a number or u[0] / u[n - 1] "
                "(examples/glow_discharge/fedm-gd.py:358)")
    if me_old is None:
        _no("no coefficient depends on the mean energy: use the LFA family")
    # ---- Poisson source: sum_i sign_i e exp(u_i) / eps0 ------------------------------------------
    charges = [0.0] * ns
    for term in forms.RateSum.coerce(poissons[0].f).terms:
        if len(term.powers) != 1 or list(term.powers.values()) != [1] or not term.coef.is_const():
            _no("the Poisson source must be sum_i sign_i e exp(u_i) / epsilon")
        i = next(iter(term.powers))
        charges[i] = term.coef.const_value() * eps0 / q_e
    for i in range(1, ns):
        if eq_type[i] == "drift-diffusion-reaction":
            _close(charges[i], sign[i], f"charge number of species {i} in the Poisson source")
        else:
            sign[i] = float(round(charges[i])) if abs(charges[i] - round(charges[i])) < 1e-9 else charges[i]
    # ---- walls -----------------------------------------------------------------------------------
    n_tags = max([b.tag for b in bterms], default=0)
    tags_mf = next((b.ds.subdomain_data for b in bterms if b.ds.subdomain_data is not None), None)
    if tags_mf is not None:
        n_tags = max(n_tags, int(np.max(tags_mf)))
    ref = [[1.0] * ns for _ in range(n_tags)]
    gamma = [0.0] * n_tags
    vth = [0.0] * ns
    is_ion = [False] * ns
    we_secondary, vth_e_coef, me_now = None, None, None
    gamma_density, gamma_energy = {}, {}
    ion_fluxes = None
    for b in bterms:
        if b.kind != "flux source":
            _no("LMEA walls are 'flux source' boundaries")
        comp = b.u.index
        sp = ie if comp == 0 else comp
        t = b.tag - 1
        ref[t][sp] = float(b.ref)
        f_mu, _, s_mu = coefficient(b.mu)
        if f_mu is not mu_rows[sp]:
            _no(f"boundary term of component {comp} must use the mobility of species {sp}")
        _close(s_mu, 5.0 / 3.0 if comp == 0 else 1.0, f"factor of the mobility in the boundary term of component {comp}")
        _close(float(b.sign), sign[sp], f"sign in the boundary term of component {comp}")
        if sp == ie:
            # thermal velocity sqrt(16 e mean_energy / (3 pi m_e)) [x 1.3333 for the energy], fedm-gd.py:224,382
            fn = [f for f in (b.vth.leaves(Function) if isinstance(b.vth, Sym) else [])]
            if len(fn) != 1:
                _no("the electron thermal velocity must be sqrt(16 e mean_energy / (3 pi m_e))")
            me_now = fn[0] if me_now is None else me_now
            if fn[0] is not me_now:
                _no("one mean-energy Function must enter all electron thermal velocities")
            v1 = float(np.ravel(evaluate(b.vth, {id(me_now): np.ones(1)}))[0])
            v4 = float(np.ravel(evaluate(b.vth, {id(me_now): np.full(1, 4.0)}))[0])
            _close(v4, 2.0 * v1, "dependence of the electron thermal velocity on the mean energy (square root)")
            coef = (v1 / (1.3333 if comp == 0 else 1.0)) ** 2
            if vth_e_coef is not None:
                _close(coef, vth_e_coef, "electron thermal velocity of the energy wall term (1.3333 x that of the density)")
            vth_e_coef = coef
            if comp == 0:
                # gamma_i * Expression(mean energy of the secondary electrons), fedm-gd.py:355,382
                g = None
                if isinstance(b.gamma, Sym) and b.gamma.op == "mul":
                    nums = [a for a in b.gamma.args if isinstance(a, (int, float))]
                    exprs = [a for a in b.gamma.args if isinstance(a, forms.Expression)]
                    if len(nums) == 1 and len(exprs) == 1:
                        g = float(nums[0])
                        if g != 0.0:
                            if we_secondary is not None:
                                _close(exprs[0].value(), we_secondary, "mean energy of the secondary electrons")
                            we_secondary = exprs[0].value()
                elif isinstance(b.gamma, (int, float)) and b.gamma == 0:
                    g = 0.0
                if g is None:
                    _no("the energy wall term needs gamma = gamma_i * Expression(mean energy of the secondaries)")
                gamma_energy[t] = g
            else:
                g = float(b.gamma)
                gamma_density[t] = g
            if not isinstance(b.Ion_flux, IonFlux):
                if not (isinstance(b.Ion_flux, (int, float)) and b.Ion_flux == 0 and float(g) == 0.0):
                    _no("electron wall terms need Ion_flux = sum Max(dot(Gamma_ion, normal), 0)")
            else:
                ion_fluxes = b.Ion_flux
        else:
            vth[sp] = float(b.vth)
    for t in range(n_tags):
        gd, ge_ = gamma_density.get(t), gamma_energy.get(t)
        if gd is not None and ge_ is not None:
            _close(gd, ge_, f"secondary-emission coefficient of wall {t + 1} (density vs energy term)")
        gamma[t] = gd if gd is not None else (ge_ or 0.0)
    if ion_fluxes is not None:
        for fl in ion_fluxes.fluxes:
            is_ion[fl.u.index] = True
    if vth_e_coef is None:
        _no("no wall term of the electrons: the device model needs their thermal velocity")
    electron_mass = 16.0 * q_e / (3.0 * np.pi * vth_e_coef)
    net = (src0.G - src0.L).astype(int)
    model = GdModel(n_species=ns, N0=src0.N0, eq_type=eq_type, grad_diffusion=grad_diff, is_ion=is_ion,
                    sign=sign, vth=vth, electron_mass=electron_mass, power=src0.P.tolist(), net=net.tolist(),
                    energy_loss=loss, energy_Ei=energy_Ei, mean_energy_form=me_form, ref=ref, gamma=gamma,
                    we_secondary=we_secondary if we_secondary is not None else 0.0,
                    quadrature_degree=int(quadrature_degree),
                    axisymmetric=not isinstance(energy.r, (int, float)))
    rows = mu_rows + D_rows + mud_rows + Dd_rows + k_rows + kd_rows + [me_old, me_now, ue_old]
    return model, mesh, tags_mf, Binding(rows)
