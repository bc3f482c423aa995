"""Pins the shading half of the oracle with the reference's own known-answer tests.

Each test restates one `#[test]` of the reference (cited file:line) against the oracle's C functions. These are the
closed-form / relational properties the reference itself asserts for OpenPBR, its BRDF helpers, the area lights,
the pixel filter and the MIS heuristics; they do not depend on sampler values beyond "a decent point set".
"""
import ctypes as C
import math

import numpy as np
import pytest

import ora

L = ora.lib()
Z = (0.0, 0.0, 1.0)


def norm(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def rec_z(normal=Z, front=1):
    r = ora.HitRecord()
    r.p = ora.v3((0, 0, 0))
    r.normal = ora.v3(normal)
    r.front_face = front
    return r


def sampler(i):  # PathSampler::new(0, 0, 0, i)   openpbr.rs:1246-1251
    return ora.Sampler(L.ora_t_sampler_new(0, 0, 0, i), i)


def scatter(m, ray_dir, rec, i):
    out = ora.Scatter()
    ok = L.ora_mat_scatter(C.byref(m), ora.v3(ray_dir), C.byref(rec), sampler(i), C.byref(out))
    return out if ok else None


def mat_eval(m, ray_dir, rec, wi):
    v, pdf = ora.V3(), C.c_float()
    ok = L.ora_mat_eval(C.byref(m), ora.v3(ray_dir), C.byref(rec), ora.v3(wi), C.byref(v), C.byref(pdf))
    return (v.np(), pdf.value) if ok else None


def glass(ior=1.5, **kw):  # OpenPBR::glass  openpbr.rs:197-204
    m = ora.default_material()
    m.transmission_weight, m.specular_ior, m.specular_roughness = 1.0, ior, 0.01
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            setattr(m, k, (C.c_float * 3)(*v))
        else:
            setattr(m, k, v)
    return m


def openpbr(**kw):
    m = ora.default_material()
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            setattr(m, k, (C.c_float * 3)(*v))
        else:
            setattr(m, k, v)
    return m


def refract(v, n, eta):  # Snell: v points away from the surface, n on v's side; eta = n_i / n_t
    v, n = norm(v), norm(n)
    cos_i = float(v @ n)
    sin2_t = eta * eta * (1.0 - cos_i * cos_i)
    assert sin2_t < 1.0
    return norm(-eta * v + (eta * cos_i - math.sqrt(1.0 - sin2_t)) * n)


R_IN = norm((-0.3, 0.2, -1.0))


def _consistency(m, n, weight_cap, min_transmitted=None, tol=1e-3):
    rec = rec_z()
    transmitted = checked = 0
    for i in range(1, n + 1):
        s = scatter(m, R_IN, rec, i)
        if s is None:
            continue
        assert not s.delta
        wi = norm(s.dir.np())
        transmitted += wi[2] < 0
        ev = mat_eval(m, R_IN, rec, wi)
        assert ev is not None
        val = s.value.np()
        assert np.abs(ev[0] - val).max() < tol * (1.0 + abs(val.max()))
        assert abs(ev[1] - s.pdf) < tol * (1.0 + s.pdf)
        w = (val / s.pdf).max()
        assert np.isfinite(w) and 0.0 <= w < weight_cap
        checked += 1
    if min_transmitted is not None:
        assert transmitted > min_transmitted
    return checked


def test_eval_matches_scatter_importance():  # openpbr.rs:1259-1285
    assert _consistency(ora.default_material(), 128, 1e9) > 32


def test_glass_transmission_is_continuous_and_eval_consistent():  # openpbr.rs:1288-1355
    _consistency(glass(specular_roughness=0.25), 256, 10.0, min_transmitted=64)
    smooth, rec = glass(), rec_z()
    for i in range(257, 257 + 128):
        s = scatter(smooth, R_IN, rec, i)
        if s is None:
            continue
        _, epdf = mat_eval(smooth, R_IN, rec, norm(s.dir.np()))
        assert abs(epdf - s.pdf) < 0.05 * (1.0 + s.pdf)
        w = (s.value.np() / s.pdf).max()
        assert np.isfinite(w) and 0.0 <= w < 10.0


def test_near_smooth_refraction_matches_snell():  # openpbr.rs:1358-1384
    m, rec = glass(), rec_z()
    d = norm((0.4, 0.0, -1.0))
    expected = refract(-d, Z, 1.0 / 1.5)
    checked = 0
    for i in range(1, 129):
        s = scatter(m, d, rec, i)
        if s is not None and s.dir.z < 0:
            assert norm(s.dir.np()) @ expected > 0.995
            checked += 1
    assert checked > 32


def test_thin_walled_transmission_stays_delta():  # openpbr.rs:1387-1410
    m, rec = glass(thin_walled=1), rec_z()
    deltas = 0
    for i in range(1, 129):
        s = scatter(m, R_IN, rec, i)
        if s is not None and s.dir.z < 0:
            assert s.delta
            deltas += 1
    assert deltas > 16
    ev, _ = mat_eval(m, R_IN, rec, (0, 0, -1))
    assert (ev == 0).all()


def test_thin_wall_window_transmittance_matches_formula():  # openpbr.rs:1413-1446
    m, rec = glass(thin_walled=1), rec_z()
    d = norm((0.6, 0.0, -1.0))
    thr = L.ora_t_sample_transmission_thin(C.byref(m), ora.v3(d), C.byref(rec)).np()
    f = L.ora_fresnel_dielectric(float(-d[2]), 1.0, 1.5)
    expected = (1.0 - f) / (1.0 + f)
    assert np.abs(thr - expected).max() < 1e-5
    assert abs(2.0 * f / (1.0 + f) + expected - 1.0) < 1e-6
    thr_n = L.ora_t_sample_transmission_thin(C.byref(m), ora.v3((0, 0, -1)), C.byref(rec))
    thr_g = L.ora_t_sample_transmission_thin(C.byref(m), ora.v3(norm((6.0, 0.0, -1.0))), C.byref(rec))
    assert thr_g.x < thr_n.x


def test_thin_wall_tint_darkens_with_angle():  # openpbr.rs:1449-1477
    m, rec = glass(thin_walled=1, transmission_color=(0.4, 0.8, 0.9)), rec_z()
    n = L.ora_t_sample_transmission_thin(C.byref(m), ora.v3((0, 0, -1)), C.byref(rec))
    assert abs(n.x / n.z - 0.4 / 0.9) < 1e-4
    o = L.ora_t_sample_transmission_thin(C.byref(m), ora.v3(norm((2.0, 0.0, -1.0))), C.byref(rec))
    assert o.x / o.z < n.x / n.z


def test_thin_wall_reflection_boosted_by_internal_bounces():  # openpbr.rs:1480-1499
    thick = glass(specular_roughness=0.25)
    thin = glass(specular_roughness=0.25, thin_walled=1)
    v, l = norm((0.4, 0, 1)), norm((-0.4, 0, 1))
    r_thick = L.ora_eval_all(C.byref(thick), ora.v3(v), ora.v3(l), 1)
    r_thin = L.ora_eval_all(C.byref(thin), ora.v3(v), ora.v3(l), 1)
    assert r_thin.x > r_thick.x * 1.5


def test_fresnel_dielectric_sanity():  # openpbr.rs:1502-1513
    assert abs(L.ora_fresnel_dielectric(1.0, 1.0, 1.5) - 0.04) < 1e-3
    assert L.ora_fresnel_dielectric(0.2, 1.5, 1.0) == 1.0
    assert L.ora_fresnel_dielectric(0.01, 1.0, 1.5) > 0.9


def test_f0_from_ior():  # openpbr.rs:1516-1524
    assert abs(L.ora_f0_from_ior(1.5) - 0.04) < 1e-3
    assert abs(L.ora_f0_from_ior(1.0)) < 1e-6


def test_iso_matches_aniso_at_zero_and_reference_remap():  # openpbr.rs:1527-1532, :1961-1969
    ax, ay = C.c_float(), C.c_float()
    L.ora_roughness_to_alpha_aniso(0.4, 0.0, C.byref(ax), C.byref(ay))
    assert abs(ax.value - 0.16) < 1e-5 and abs(ay.value - 0.16) < 1e-5
    r, a = 0.5, 0.8
    L.ora_roughness_to_alpha_aniso(r, a, C.byref(ax), C.byref(ay))
    inv = 1.0 - a
    want = r * r * math.sqrt(2.0 / (1.0 + inv * inv))
    assert abs(ax.value - want) < 1e-6 and abs(ay.value - inv * want) < 1e-6


def test_sheen_nonneg():  # openpbr.rs:1535-1544
    for nv in (0.1, 0.3, 0.5, 0.9):
        for nl in (0.1, 0.3, 0.5, 0.9):
            for nh in (0.1, 0.5, 0.9):
                v = L.ora_t_sheen_charlie(nv, nl, nh, 0.4)
                assert math.isfinite(v) and v >= 0.0


def test_defaults_match_spec():  # openpbr.rs:1547-1555
    m = ora.default_material()
    assert m.base_weight == 1.0 and m.specular_ior == 1.5
    assert abs(m.coat_ior - 1.6) < 1e-7 and abs(m.thin_film_ior - 1.4) < 1e-7
    assert m.transmission_dispersion_abbe_number == 20.0
    assert tuple(m.subsurface_radius_scale) == (1.0, 0.5, 0.25)


def test_coat_darkening_identity_at_zero():  # openpbr.rs:1558-1562
    v = L.ora_t_coat_darkening_factor(ora.v3((0.7, 0.3, 0.2)), 1.6, 0.0).np()
    assert np.linalg.norm(v - 1.0) < 1e-4


def test_thin_film_bounded():  # openpbr.rs:1565-1570, :1972-1981
    r = L.ora_thin_film_fresnel(1.0, 1.0, 1.4, 1.5, 500.0).np()
    assert ((r >= 0) & (r <= 1)).all()
    f0 = (0.9, 0.7, 0.4)
    rm = L.ora_t_thin_film_fresnel_metal(0.8, 1.0, 1.4, ora.v3(f0), 500.0).np()
    assert ((rm >= 0) & (rm <= 1)).all()
    plain = L.ora_fresnel_f82_tint(0.8, ora.v3(f0), ora.v3((1, 1, 1))).np()
    assert np.abs(rm - plain).max() > 1e-3


def test_coat_lobe_contributes_when_enabled():  # openpbr.rs:1573-1598
    m = openpbr(coat_weight=1.0, coat_roughness=0.05, coat_ior=1.5, base_color=(0.5, 0.5, 0.5))
    rec = rec_z(front=0)
    d = norm((-0.5, 0.0, -1.0))
    assert any((s := scatter(m, d, rec, i)) is not None and (s.value.np() ** 2).sum() > 0 for i in range(1, 129))


def test_dispersive_ior_properties():  # openpbr.rs:1601-1660
    v = L.ora_dispersive_ior(1.5, 30.0, 0.0)
    assert (v.x, v.y, v.z) == (1.5, 1.5, 1.5)
    v = L.ora_dispersive_ior(1.5, 30.0, 1.0)
    assert v.z > v.y > v.x
    n_d, v_d = 1.5, 30.0
    assert abs(L.ora_cauchy_ior(n_d, v_d, 587.6) - n_d) < 1e-6
    abbe = (n_d - 1.0) / (L.ora_cauchy_ior(n_d, v_d, 486.1) - L.ora_cauchy_ior(n_d, v_d, 656.3))
    assert abs(abbe - v_d) < 0.05
    full, half = L.ora_dispersive_ior(1.5, 40.0, 1.0), L.ora_dispersive_ior(1.5, 40.0, 0.5)
    assert abs((full.z - full.x) / (half.z - half.x) - 2.0) < 1e-3
    assert abs(full.y - 1.5) < 0.01
    n, inv = L.ora_dispersive_ior(1.5, 30.0, 1.0).np(), L.ora_dispersive_ior(1.0 / 1.5, 30.0, 1.0).np()
    assert np.abs(inv - 1.0 / n).max() < 1e-5
    assert inv[2] < inv[1] < inv[0]


def test_dispersive_glass_is_continuous_and_eval_consistent():  # openpbr.rs:1663-1708
    _consistency(glass(specular_roughness=0.25, transmission_dispersion_scale=1.0), 256, 30.0, min_transmitted=64)


def test_dispersion_separates_channels():  # openpbr.rs:1711-1735
    m, rec = glass(transmission_dispersion_scale=1.0), rec_z()
    d = norm((0.6, 0.0, -1.0))
    eta_g = L.ora_t_transmission_iors(C.byref(m)).y
    v, pdf = mat_eval(m, d, rec, refract(-d, Z, 1.0 / eta_g))
    assert pdf > 0 and v[1] > 0 and v[1] > v[0] and v[1] > v[2]


def test_eon_properties():  # openpbr.rs:1738-1775
    rho = (0.8, 0.5, 0.3)
    f = L.ora_eon_diffuse(ora.v3(rho), 0.0, ora.v3(norm((0.3, 0.1, 0.9))), ora.v3(norm((-0.2, 0.4, 0.8)))).np()
    assert np.abs(f - np.array(rho) / math.pi).max() < 1e-4
    v, l = ora.v3(norm((0.5, -0.1, 0.6))), ora.v3(norm((-0.3, 0.2, 0.9)))
    for r in (0.2, 0.6, 1.0):
        a = L.ora_eon_diffuse(ora.v3((0.9, 0.6, 0.2)), r, v, l).np()
        b = L.ora_eon_diffuse(ora.v3((0.9, 0.6, 0.2)), r, l, v).np()
        assert np.abs(a - b).max() < 1e-5
    for i in range(1, 21):
        for r in (0.0, 0.3, 0.7, 1.0):
            assert abs(L.ora_eon_albedo_exact(i / 20.0, r) - L.ora_eon_albedo_approx(i / 20.0, r)) < 0.01


def test_eon_preserves_energy_at_high_roughness():  # openpbr.rs:1778-1805 (white furnace, quadrature 64x64)
    n_t = n_p = 64
    one = ora.v3((1, 1, 1))
    for view_z in (0.95, 0.6, 0.25):
        v = ora.v3((math.sqrt(1.0 - view_z * view_z), 0.0, view_z))
        integral = 0.0
        for it in range(n_t):
            theta = (it + 0.5) / n_t * (math.pi / 2)
            st, ct = math.sin(theta), math.cos(theta)
            for ip in range(n_p):
                phi = (ip + 0.5) / n_p * 2 * math.pi
                f = L.ora_eon_diffuse(one, 1.0, v, ora.V3(st * math.cos(phi), st * math.sin(phi), ct)).x
                integral += f * ct * st
        integral *= (math.pi / 2) / n_t * (2 * math.pi) / n_p
        assert 0.97 <= integral <= 1.03, (view_z, integral)


def test_deep_transmission_interface_is_untinted():  # openpbr.rs:1808-1844
    color = (0.9, 0.4, 0.2)
    shallow = glass(specular_roughness=0.25, transmission_color=color)
    deep = glass(specular_roughness=0.25, transmission_color=color, transmission_depth=1.0)
    rec = rec_z()
    d = norm((0.4, 0.0, -1.0))
    wi = refract(-d, Z, 1.0 / 1.5)
    vd, _ = mat_eval(deep, d, rec, wi)
    assert vd[1] > 0 and abs(vd[0] - vd[1]) < 1e-5 and abs(vd[1] - vd[2]) < 1e-5
    vs, _ = mat_eval(shallow, d, rec, wi)
    assert abs(vs[0] / vs[1] - color[0] / color[1]) < 1e-3


def test_f82_metal_edge_tint():  # openpbr.rs:1847-1863
    f0, tint, one = ora.v3((0.9, 0.6, 0.3)), ora.v3((1.0, 0.5, 0.25)), ora.v3((1, 1, 1))
    assert np.abs(L.ora_fresnel_f82_tint(1.0, f0, tint).np() - f0.np()).max() < 1e-5
    assert np.abs(L.ora_fresnel_f82_tint(0.0, f0, tint).np() - 1.0).max() < 1e-5
    w, wo = L.ora_fresnel_f82_tint(1.0 / 7.0, f0, tint).np(), L.ora_fresnel_f82_tint(1.0 / 7.0, f0, one).np()
    assert abs(w[1] / wo[1] - 0.5) < 1e-3 and abs(w[2] / wo[2] - 0.25) < 1e-3 and abs(w[0] - wo[0]) < 1e-5


def test_coat_color_tints_substrate_not_coat_reflection():  # openpbr.rs:1866-1888
    m = openpbr(coat_weight=1.0, coat_color=(0.9, 0.2, 0.2), coat_darkening=0.0)
    v = ora.v3(norm((0.3, 0.0, 1.0)))
    c = L.ora_t_eval_coat(C.byref(m), v, v, v, 0.1, 0.1)
    assert abs(c.x - c.y) < 1e-6 and abs(c.y - c.z) < 1e-6
    a = L.ora_t_coat_attenuation(C.byref(m), 1.0, 1.0)
    assert abs(a.x / a.y - 0.9 / 0.2) < 1e-3


def test_coat_round_trip_recovers_authored_color():  # openpbr.rs:1891-1907
    cc = (0.9, 0.4, 0.16)
    m = openpbr(coat_weight=1.0, coat_color=cc, coat_darkening=0.0)
    a = L.ora_t_coat_attenuation(C.byref(m), 1.0, 1.0).np()
    f0 = L.ora_f0_from_ior(m.coat_ior)
    assert np.abs(a - np.array(cc) * (1 - f0) * (1 - f0)).max() < 1e-4


def test_coat_passage_darkens_and_saturates_at_grazing():  # openpbr.rs:1910-1930
    m = openpbr(coat_weight=1.0, coat_color=(0.9, 0.3, 0.3))
    pn, pg = L.ora_coat_passage(C.byref(m), 1.0), L.ora_coat_passage(C.byref(m), 0.2)
    assert pg.x < pn.x and pg.y < pn.y and pg.y / pg.x < pn.y / pn.x
    assert L.ora_t_coat_attenuation(C.byref(m), 0.2, 0.2).y < L.ora_t_coat_attenuation(C.byref(m), 1.0, 1.0).y


def test_coated_emission_dims_and_tints():  # openpbr.rs:1933-1958
    un = openpbr(emission_luminance=100.0)
    assert (L.ora_mat_emitted_directional(C.byref(un), 0.3).np() == L.ora_mat_emitted(C.byref(un)).np()).all()
    co = openpbr(emission_luminance=100.0, coat_weight=1.0, coat_color=(1.0, 0.2, 0.2))
    en = L.ora_mat_emitted_directional(C.byref(co), 1.0)
    assert en.x < 100.0 and abs(en.y / en.x - math.sqrt(0.2)) < 1e-3
    assert L.ora_mat_emitted_directional(C.byref(co), 0.05).x < en.x


def test_subsurface_shifts_diffuse_color():  # openpbr.rs:2134-2171
    sss = openpbr(base_color=(0.9, 0.9, 0.9), subsurface_color=(0.9, 0.1, 0.1), subsurface_weight=1.0,
                  base_diffuse_roughness=0.5)
    no = openpbr(base_color=(0.9, 0.9, 0.9), base_diffuse_roughness=0.5)
    rec = rec_z()
    a, b = np.zeros(3), np.zeros(3)
    k = 0
    for _ in range(512):
        k += 1
        s = scatter(sss, (0, 0, -1), rec, k)
        if s is not None:
            a += s.value.np()
        k += 1
        s = scatter(no, (0, 0, -1), rec, k)
        if s is not None:
            b += s.value.np()
    assert a[1] < b[1] and a[2] < b[2]


# ---- carried interior medium: openpbr.rs:1985-2131, medium.rs:186-245 ----
def _interior(m):
    """sample_interior_medium (openpbr.rs:1985-2000): the medium of the first sample that refracts into the surface."""
    rec = rec_z()
    for i in range(1, 257):
        s = scatter(m, R_IN, rec, i)
        if s is not None and s.dir.z < 0:
            med = ora.Medium()
            has = L.ora_interior_medium(C.byref(m), C.byref(med))
            assert bool(s.medium) == bool(has)
            return med if s.medium else None
    raise AssertionError("material never transmitted")


def test_transmission_scatter_wires_into_interior_medium():  # openpbr.rs:2003-2030
    m = glass(specular_roughness=0.25, transmission_color=(0.5, 0.7, 0.9), transmission_depth=2.0,
              transmission_scatter=(0.2, 0.2, 0.1), transmission_scatter_anisotropy=0.5)
    med = _interior(m)
    assert med is not None and L.ora_medium_is_scattering(C.byref(med))
    assert np.abs(med.sigma_s.np() - np.array([0.1, 0.1, 0.05])).max() < 1e-5
    ext = med.sigma_a.np() + med.sigma_s.np()
    assert np.abs(ext - (-np.log(np.array([0.5, 0.7, 0.9])) / 2.0)).max() < 1e-5
    assert med.g == 0.5


def test_transmission_scatter_negative_absorption_shifts_to_gray():  # openpbr.rs:2034-2043
    m = L.ora_medium_from_transmission(ora.v3((1, 1, 1)), 1.0, ora.v3((0.1, 0.2, 0.4)), 0.0)
    a = m.sigma_a.np()
    assert a.min() >= 0.0 and abs(a[2]) < 1e-6 and a[0] > a[1] > a[2]


def test_van_de_hulst_inversion_boosts_single_scatter_albedo():  # openpbr.rs:2046-2066
    one = ora.v3((1, 1, 1))
    m = L.ora_medium_from_subsurface(ora.v3((0.5,) * 3), 1.0, one, 0.0)
    a = L.ora_medium_albedo(C.byref(m)).x
    assert abs(a - 0.9117) < 5e-3
    hi = L.ora_medium_from_subsurface(ora.v3((0.95,) * 3), 1.0, one, 0.0)
    lo = L.ora_medium_from_subsurface(ora.v3((0.05,) * 3), 1.0, one, 0.0)
    assert L.ora_medium_albedo(C.byref(hi)).x > 0.99
    assert L.ora_medium_albedo(C.byref(lo)).x < a < L.ora_medium_albedo(C.byref(hi)).x
    fwd = L.ora_medium_from_subsurface(ora.v3((0.5,) * 3), 1.0, one, 0.9)
    assert L.ora_medium_albedo(C.byref(fwd)).x > a
    assert abs((m.sigma_a.x + m.sigma_s.x) - 1.0) < 1e-5


def test_interior_medium_blends_subsurface_into_glass():  # openpbr.rs:2069-2098
    m = glass(specular_roughness=0.25, transmission_depth=1.0, subsurface_weight=1.0, subsurface_color=(0.5, 0.5, 0.5),
              subsurface_radius=0.1, subsurface_radius_scale=(1.0, 1.0, 1.0), subsurface_scatter_anisotropy=0.3,
              transmission_weight=0.5)
    med = _interior(m)
    assert med is not None and L.ora_medium_is_scattering(C.byref(med))
    assert abs(med.g - 0.3) < 1e-5
    full = L.ora_medium_from_subsurface(ora.v3((0.5,) * 3), 0.1, ora.v3((1, 1, 1)), 0.3)
    assert np.abs(med.sigma_s.np() - full.sigma_s.np() * 0.5).max() < 1e-3


def test_inert_glass_attaches_no_medium():  # openpbr.rs:2102-2113
    assert _interior(glass(specular_roughness=0.25)) is None


def test_medium_transmittance():  # openpbr.rs:2116-2131
    m0 = L.ora_medium_from_transmission(ora.v3((0.5, 0.5, 0.5)), 0.0, ora.v3((0, 0, 0)), 0.0)
    assert np.linalg.norm(L.ora_medium_transmittance(C.byref(m0), 1.0).np() - 1.0) < 1e-4
    m = L.ora_medium_from_transmission(ora.v3((0.5, 0.7, 0.9)), 1.0, ora.v3((0, 0, 0)), 0.0)
    short, long_ = L.ora_medium_transmittance(C.byref(m), 0.1).np(), L.ora_medium_transmittance(C.byref(m), 2.0).np()
    assert (long_ < short).all()
    # at the authored depth the transmittance is the authored colour (medium.rs:8-11)
    assert np.abs(L.ora_medium_transmittance(C.byref(m), 1.0).np() - np.array([0.5, 0.7, 0.9])).max() < 1e-5


def test_hg_phase_normalizes_over_sphere():  # medium.rs:190-206
    for g in (-0.7, 0.0, 0.4, 0.9):
        n = 20000
        mu = -1.0 + 2.0 * (np.arange(n) + 0.5) / n
        total = sum(L.ora_hg_phase(float(x), g) for x in mu)
        assert abs(2.0 * math.pi * total * (2.0 / n) - 1.0) < 1e-3, g


def test_hg_sampling_matches_the_phase_convention():  # medium.rs:208-245: g > 0 scatters forward, isotropic at g = 0
    wi = ora.v3((0, 0, 1))
    s = np.zeros(3)
    for i in range(2048):  # openpbr.rs:2174-2188
        s += L.ora_sample_henyey_greenstein(wi, 0.0, ((i * 13 + 7) % 1024) / 1024.0, ((i * 31 + 5) % 1024) / 1024.0).np()
    assert np.linalg.norm(s / 2048.0) < 0.05
    for g in (0.6, -0.6):
        mean = np.mean([L.ora_sample_henyey_greenstein(wi, g, (i + 0.5) / 512.0, ((i * 31 + 5) % 512) / 512.0).z
                        for i in range(512)])
        assert abs(mean - g) < 0.02  # E[cos theta] = g for Henyey-Greenstein
        d = L.ora_sample_henyey_greenstein(ora.v3(norm((0.3, -0.5, 0.8))), g, 0.37, 0.81).np()
        assert abs(np.linalg.norm(d) - 1.0) < 1e-5


def test_thin_walled_transmission_scatters_downward():  # openpbr.rs:2191-2216
    m = openpbr(transmission_weight=1.0, transmission_color=(0.7, 0.9, 0.7), thin_walled=1)
    rec = rec_z(normal=(0, 1, 0))
    got = False
    for i in range(1, 65):
        s = scatter(m, (0, -1, 0), rec, i)
        if s is not None and s.dir.y < 0:
            assert s.delta
            got = True
            break
    assert got


def test_scatter_importance_finite():  # openpbr.rs:2219-2244
    m = openpbr(base_color=(0.7, 0.3, 0.2), base_metalness=0.3, fuzz_weight=0.2, fuzz_color=(0.9, 0.9, 0.9))
    rec = rec_z(normal=(0, 1, 0), front=0)
    for i in range(1, 65):
        s = scatter(m, norm((0, -1, -1)), rec, i)
        if s is not None:
            assert math.isfinite(s.pdf) and s.pdf > 0
            v = s.value.np()
            assert np.isfinite(v).all() and (v >= 0).all()


# ---- light.rs ----
def sphere_light(center, radius, radiance=(10, 10, 10), geom_id=0):
    l = ora.Light()
    l.kind, l.geom_id, l.radius = ora.LIGHT_SPHERE, geom_id, radius
    l.center = (C.c_float * 3)(*center)
    l.radiance = (C.c_float * 3)(*radiance)
    return l


def test_sphere_shape_samples_lie_on_surface():  # light.rs:453-465
    l = sphere_light((1, 2, 3), 0.5)
    for u, v in [(0.0, 0.0), (0.25, 0.75), (0.99, 0.5), (0.5, 0.01)]:
        p = L.ora_t_light_sample_point(C.byref(l), u, v)
        assert abs(np.linalg.norm(p.np() - np.array([1, 2, 3])) - 0.5) < 1e-5
        assert abs(np.linalg.norm(L.ora_t_light_normal_at(C.byref(l), p).np()) - 1.0) < 1e-5


def test_rect_shape_samples_lie_in_rect():  # light.rs:468-479
    l = ora.Light()
    l.kind = ora.LIGHT_RECT
    l.origin, l.edge_u = (C.c_float * 3)(-1, 5, -2), (C.c_float * 3)(2, 0, 0)
    l.edge_v, l.normal = (C.c_float * 3)(0, 0, 4), (C.c_float * 3)(0, -1, 0)
    assert abs(L.ora_t_light_area(C.byref(l)) - 8.0) < 1e-5
    p = L.ora_t_light_sample_point(C.byref(l), 0.5, 0.5)
    assert np.linalg.norm(p.np() - np.array([0, 5, 0])) < 1e-5
    assert (L.ora_t_light_normal_at(C.byref(l), p).np() == np.array([0, -1, 0], dtype=np.float32)).all()


def distant(direction, irradiance, angle):
    l = ora.Light()
    L.ora_light_distant(C.byref(l), ora.v3(direction), ora.v3(irradiance), angle)
    return l


def escaped(l, direction):
    rad, pdf = ora.V3(), C.c_float()
    if not L.ora_light_escaped(C.byref(l), ora.v3(direction), C.byref(rad), C.byref(pdf)):
        return None
    return rad.np(), pdf.value


def off_axis(axis, degrees):
    a = math.radians(degrees)
    d = L.ora_align_to_normal(ora.v3((math.sin(a), 0.0, math.cos(a))), ora.v3(axis)).np()
    return d / np.linalg.norm(d)


def test_distant_light_cone_is_consistent():  # light.rs:527-566
    d = np.array([0.3, -1.0, 0.2]) / np.linalg.norm([0.3, -1.0, 0.2])
    l = distant(d, (2, 2, 2), 10.0)
    rng = np.random.default_rng(7)
    s = ora.LightSample()
    for _ in range(2000):
        assert L.ora_light_sample_li(C.byref(l), ora.v3((0, 0, 0)), rng.random(), rng.random(), C.byref(s))
        assert abs(np.linalg.norm(s.direction.np()) - 1.0) < 1e-5
        assert math.isinf(s.distance)
        e = escaped(l, s.direction.np())
        assert e is not None, "sample_li produced a direction escaped() does not cover"
        assert (e[0] == s.radiance.np()).all()
        assert abs(e[1] - s.pdf) < 1e-3 * s.pdf
    assert escaped(l, d) is None
    assert escaped(l, off_axis(-np.asarray(l.normal[:]), 20.0)) is None


def test_distant_light_irradiance_is_angle_invariant():  # light.rs:576-591
    e = np.array([3.0, 2.0, 1.0])
    s = ora.LightSample()
    for angle in (0.0, 0.53, 5.0, 30.0):
        l = distant((0, -1, 0), e, angle)
        assert L.ora_light_sample_li(C.byref(l), ora.v3((0, 0, 0)), 0.4, 0.6, C.byref(s))
        recovered = s.radiance.np() / s.pdf
        assert np.linalg.norm(recovered - e) < 1e-3 * np.linalg.norm(e), angle


def test_distant_light_zero_angle_stays_finite():  # light.rs:595-612
    l = distant((0, -1, 0), (1, 1, 1), 0.0)
    s = ora.LightSample()
    assert L.ora_light_sample_li(C.byref(l), ora.v3((0, 0, 0)), 0.5, 0.5, C.byref(s))
    assert math.isfinite(s.pdf) and s.pdf > 0
    assert np.isfinite(s.radiance.np()).all()
    assert escaped(l, off_axis((0, 1, 0), 1.0)) is None


def test_distant_light_has_no_geometry():  # light.rs:616-621
    l = distant((0, -1, 0), (1, 1, 1), 1.0)
    assert l.geom_id == 0xFFFFFFFF
    assert L.ora_light_pdf_at_point(C.byref(l), ora.v3((0, 0, 0)), ora.v3((0, 1, 0))) == 0.0


def test_distant_light_python_form_matches_oracle_constructor():
    """The derived fields the importer hands to both renderers follow DistantLight::new (light.rs:255-266)."""
    import importlib
    usda = importlib.import_module("crust-render_amd.usda")
    for direction, angle in (((0.3, -1.0, 0.2), 10.0), ((0, -1, 0), 0.53), ((1, 1, 1), 0.0), ((0, 0, -2), 400.0)):
        d = usda.distant_light(direction, (1, 2, 3), angle)
        l = distant(direction, (1, 2, 3), angle)
        assert np.allclose(d["direction"], l.normal[:], rtol=0, atol=1.2e-7)
        assert abs(d["cos_half_angle"] - l.radius) <= 1.2e-7
        assert abs(d["solid_angle"] - l.center[0]) <= 1e-6 * max(l.center[0], 1e-6) + 8e-7
    assert usda.distant_light((0, 0, 0), (1, 1, 1)) is None


def test_uniform_dome_covers_every_direction():  # light.rs:340-390 without a map
    l = ora.Light()
    l.kind, l.geom_id = ora.LIGHT_DOME, 0xFFFFFFFF
    l.radiance = (C.c_float * 3)(0.5, 0.6, 0.7)
    rng = np.random.default_rng(3)
    s = ora.LightSample()
    mean = np.zeros(3)
    for _ in range(2000):
        assert L.ora_light_sample_li(C.byref(l), ora.v3((0, 0, 0)), rng.random(), rng.random(), C.byref(s))
        assert abs(np.linalg.norm(s.direction.np()) - 1.0) < 1e-5 and math.isinf(s.distance)
        assert abs(s.pdf - 1 / (4 * math.pi)) < 1e-7
        e = escaped(l, s.direction.np())
        assert e is not None and (e[0] == s.radiance.np()).all() and e[1] == s.pdf
        mean += s.direction.np()
    assert np.linalg.norm(mean / 2000) < 0.05  # uniform over the sphere


def test_area_light_pdf_is_positive_facing_side():  # light.rs:482-521
    l = sphere_light((0, 5, 0), 1.0)
    pdf = L.ora_light_pdf_at_point(C.byref(l), ora.v3((0, 0, 0)), ora.v3((0, 4, 0)))
    assert math.isfinite(pdf) and pdf > 0
    s = ora.LightSample()
    assert L.ora_light_sample_li(C.byref(l), ora.v3((0, 0, 0)), 0.3, 0.7, C.byref(s))
    assert abs(np.linalg.norm(s.direction.np()) - 1.0) < 1e-5
    assert math.isfinite(s.distance) and s.distance > 0
    assert (s.radiance.np() == 10.0).all()
    assert math.isfinite(s.pdf) and s.pdf > 0
    point = s.direction.np().astype(np.float64) * s.distance
    fp = L.ora_light_pdf_at_point(C.byref(l), ora.v3((0, 0, 0)), ora.v3(point))
    assert abs(s.pdf - fp) <= 1e-3 * max(s.pdf, fp)


# ---- filter.rs ----
def _filt(kind, r, u):
    x, w = C.c_float(), C.c_float()
    L.ora_filter_sample(kind, r, u, C.byref(x), C.byref(w))
    return x.value, w.value


def test_box_half_radius_is_the_identity_jitter():  # filter.rs:272-282
    rng = np.random.default_rng(7)
    for u in rng.random(10000, dtype=np.float32):
        x, w = _filt(0, 0.5, float(u))
        assert np.float32(x).tobytes() == np.float32(u).tobytes() and w == 1.0


def test_filter_samples_stay_in_support_and_monotone():  # filter.rs:286-305 (box, triangle)
    for kind, r in ((0, 0.5), (1, 1.0), (1, 1.5)):
        prev = -math.inf
        for i in range(1001):
            x, _ = _filt(kind, r, min(i / 1000.0, 0.999999))
            assert 0.5 - r - 1e-3 <= x <= 0.5 + r + 1e-3 and x >= prev
            prev = x


def test_triangle_histogram_reproduces_the_kernel():  # filter.rs:340-371
    r, H, n = 1.0, 16, 100000
    rng = np.random.default_rng(23)
    hist = np.zeros(H)
    for u in rng.random(n, dtype=np.float32):
        x, w = _filt(1, r, float(u))
        hist[min(int(((x - 0.5 + r) / (2 * r)) * H), H - 1)] += w / n
    mids = -r + (np.arange(H) + 0.5) * (2 * r / H)
    want = np.maximum(1.0 - np.abs(mids) / r, 0.0)
    want /= want.sum()
    assert np.abs(hist - want).max() < 0.01


# ---- common.rs ----
def test_mis_heuristics():  # common.rs:38-49, tracer.rs:85-104
    assert abs(L.ora_balance_heuristic(1.0, 1.0) - 0.5) < 1e-6
    assert abs(L.ora_power_heuristic(1.0, 1.0) - 0.5) < 1e-6
    assert abs(L.ora_power_heuristic(2.0, 1.0) - 0.8) < 1e-6
    for a, b in [(0.3, 2.0), (5.0, 0.01), (1e-3, 1e-3)]:
        for f in (L.ora_balance_heuristic, L.ora_power_heuristic):
            assert abs(f(a, b) + f(b, a) - 1.0) < 1e-3 + 1e-6 / (a * a + b * b)


# ---- sampler: the properties the reference's tests rely on ("parity unpinned" for values) ----
def test_sampler_is_stratified_and_domains_decorrelate():
    pat = L.ora_t_sampler_new(3, 5, 0, 0)
    pts = np.zeros((256, 4), dtype=np.float32)
    for i in range(256):
        L.ora_t_draw4(pat, i, pts[i].ctypes.data_as(C.POINTER(C.c_float)))
    assert ((pts >= 0) & (pts < 1)).all()
    for d in range(4):  # Owen-scrambled Sobol: each 1-D projection of 256 points hits every 1/256 stratum once
        assert len(set((pts[:, d] * 256).astype(int))) == 256
    for a, b in ((0, 1), (0, 2), (0, 3), (1, 2)):  # (0,2)-net pairs of Joe-Kuo dims 0-3: one point per 16x16 stratum
        cells = set(zip((pts[:, a] * 16).astype(int), (pts[:, b] * 16).astype(int)))
        assert len(cells) == 256
    other = L.ora_t_new_domain(pat, 7)
    assert other != pat and L.ora_t_new_domain(pat, 7) == other and L.ora_t_new_domain(pat, 8) != other
    q = np.zeros(4, dtype=np.float32)
    L.ora_t_draw4(other, 0, q.ctypes.data_as(C.POINTER(C.c_float)))
    assert not np.array_equal(q, pts[0])
    assert L.ora_t_sampler_new(3, 5, 0, 0) != L.ora_t_sampler_new(4, 5, 0, 0) != L.ora_t_sampler_new(3, 5, 1, 0)
    r = [L.ora_t_rnd1(pat, i) for i in range(2048)]
    assert 0.45 < np.mean(r) < 0.55 and min(r) >= 0 and max(r) < 1


# ---- tracer.rs:1615-1668: the sampling strategies' MIS weights ----
_STRATEGIES = {"power": 0, "balance": 1, "light": 2, "bsdf": 3}  # ORA_STRATEGY_* (tracer.rs:63-75)


def test_strategy_weights_partition_unity():  # tracer.rs:1620-1647
    for s in _STRATEGIES.values():
        for light_pdf, bounce_pdf in [(0.5, 0.5), (1e-4, 1e4), (1e4, 1e-4), (3.0, 0.2), (0.05, 40.0)]:
            total = L.ora_light_weight(s, light_pdf, bounce_pdf) + L.ora_bounce_weight(s, bounce_pdf, light_pdf)
            assert abs(total - 1.0) < 1e-3, (s, light_pdf, bounce_pdf, total)


def test_single_strategy_modes_disable_the_other_side():  # tracer.rs:1649-1657
    assert L.ora_light_weight(_STRATEGIES["light"], 1.0, 100.0) == 1.0
    assert L.ora_bounce_weight(_STRATEGIES["light"], 100.0, 1.0) == 0.0
    assert L.ora_light_weight(_STRATEGIES["bsdf"], 100.0, 1.0) == 0.0
    assert L.ora_bounce_weight(_STRATEGIES["bsdf"], 1.0, 100.0) == 1.0


def test_power_sharpens_balance():  # tracer.rs:1659-1667
    a, b = 10.0, 1.0
    assert L.ora_light_weight(_STRATEGIES["power"], a, b) > L.ora_light_weight(_STRATEGIES["balance"], a, b)
