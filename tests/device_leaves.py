"""A device-backed stand-in for the oracle's ctypes library (oracle_py.load()) in the leaf tests: every `orc_fn_<name>(...)` it implements takes the SAME arguments as
the oracle's test entry point of that name and evaluates the function ON THE GPU through the TEST library's entry shm_debug_eval_leaf (libshimmer_hip_probe.so: shimmer_amd/csrc/shm/probe.h,
csrc/probe/k_leaf_probe.hip, include/shimmer_hip_probe.h — the same shared arithmetic headers the render kernels are compiled from). tests/test_gpu_leaf_replay.py calls the CPU tests' own bodies (tests/test_oracle_golden.py, test_leaf_golden.py, test_layered_golden.py) with this
object in the oracle library's place: the committed golden vectors — the reference's in-source known answers and the independent re-evaluations — meet the device code
directly, at the CPU tests' tolerances."""
import ctypes as C
import struct

import numpy as np

from shimmer_amd import abi

OPS = ("next_float_up next_float_down gamma difference_of_products dot cross coordinate_system hypot round atan2 intersect_p_cached intersect_triangle tr_d tr_g tr_lambda "
       "tr_sample_wm fresnel_dielectric fresnel_complex bxdf_sample_f bxdf_f_pdf layered_f_pdf layered_sample_f offset_ray_origin triangle_sample_with_context "
       "triangle_pdf_with_context triangle_interaction sphere_sample_with_context sphere_pdf_with_context area_light_l film_add_sample camera_ray_differential interval_op det3 "
       "rotate_from_to sample_discrete sampler_stream sample_visible_wavelengths visible_wavelengths_pdf vecmath transform_apply blp_intersect blp_sample_with_context "
       "blp_pdf_with_context sphere_intersect unary equal_area_square_to_sphere equal_area_sphere_to_square").split()
OP = {name: i + 1 for i, name in enumerate(OPS)}  # (shm/probe.h: PROBE_* in this order, from 1)


def _f(x):
    """the bits of x as a float32"""
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def _fl(seq, n=None):
    """words of a float sequence (a ctypes array, a pointer to one with n given, a list)"""
    if seq is None:
        return [0] * (n or 0)
    vals = [seq[i] for i in range(n)] if n is not None else list(seq)
    return [_f(v) for v in vals]


def _i(x):
    return int(x) & 0xFFFFFFFF


def _struct_words(obj):
    raw = bytes(obj)
    assert len(raw) % 4 == 0
    return list(struct.unpack(f"<{len(raw) // 4}I", raw))


def _deref(p):
    """a ctypes byref() / pointer argument -> the object"""
    return p._obj if hasattr(p, "_obj") else p.contents


class DeviceLeaves:
    def __init__(self, lib, device=0):
        self.lib, self.device = lib, device  # (lib: the product library — the scene-building entries some leaf tests use)
        self.probe = abi.load_probe_library()

    def _run(self, name, words, n_out):
        inp = (C.c_uint32 * len(words))(*words)
        out = (C.c_uint32 * n_out)()
        res = C.c_int(0)
        rc = self.probe.shm_debug_eval_leaf(self.device, OP[name], inp, len(words), out, n_out, C.byref(res))
        if rc != abi.SHM_OK:
            raise abi.ShimmerHipError(f"shm_debug_eval_leaf {name} failed ({rc}): {self.probe.shm_probe_last_error().decode()}")
        return res.value, np.frombuffer(bytes(out), dtype=np.uint32).copy()

    def _scalar(self, name, words):
        return float(self._run(name, words, 1)[1].view(np.float32)[0])

    @staticmethod
    def _store(dst, words, n=None):
        f = words.view(np.float32)
        for k in range(len(f) if n is None else n):
            dst[k] = float(f[k])

    # ---- float.rs / math.rs / vecmath ----
    def orc_fn_next_float_up(self, v): return self._scalar("next_float_up", [_f(v)])
    def orc_fn_next_float_down(self, v): return self._scalar("next_float_down", [_f(v)])
    def orc_fn_gamma(self, n): return self._scalar("gamma", [_i(n)])
    def orc_fn_difference_of_products(self, a, b, c, d): return self._scalar("difference_of_products", [_f(a), _f(b), _f(c), _f(d)])
    def orc_fn_dot(self, a, b): return self._scalar("dot", _fl(a, 3) + _fl(b, 3))
    def orc_fn_hypot(self, x, y): return self._scalar("hypot", [_f(x), _f(y)])

    def orc_fn_equal_area_square_to_sphere(self, uv, out3):
        _, o = self._run("equal_area_square_to_sphere", _fl(uv, 2), 3)
        out3[:] = list(o.view(np.float32))

    def orc_fn_equal_area_sphere_to_square(self, d, out2):
        _, o = self._run("equal_area_sphere_to_square", _fl(d, 3), 2)
        out2[:] = list(o.view(np.float32))
    def orc_fn_round(self, x): return self._scalar("round", [_f(x)])
    def orc_fn_atan2(self, y, x): return self._scalar("atan2", [_f(y), _f(x)])
    def orc_fn_det3(self, m): return self._scalar("det3", _fl(m, 9))
    def orc_fn_sample_visible_wavelengths(self, u): return self._scalar("sample_visible_wavelengths", [_f(u)])
    def orc_fn_visible_wavelengths_pdf(self, l): return self._scalar("visible_wavelengths_pdf", [_f(l)])

    def orc_fn_cross(self, a, b, out):
        self._store(out, self._run("cross", _fl(a, 3) + _fl(b, 3), 3)[1])

    def orc_fn_coordinate_system(self, v, out6):
        self._store(out6, self._run("coordinate_system", _fl(v, 3), 6)[1])

    def orc_fn_vecmath(self, a, b, out9):
        self._store(out9, self._run("vecmath", _fl(a, 3) + _fl(b, 3), 9)[1])

    def orc_fn_rotate_from_to(self, frm, to, v, out3):
        self._store(out3, self._run("rotate_from_to", _fl(frm, 3) + _fl(to, 3) + _fl(v, 3), 3)[1])

    def orc_fn_interval_op(self, op, alo, ahi, blo, bhi, out2):
        self._store(out2, self._run("interval_op", [_i(op), _f(alo), _f(ahi), _f(blo), _f(bhi)], 2)[1])

    def orc_fn_transform_apply(self, kind, inverse, m, m_inv, v, out3):
        self._store(out3, self._run("transform_apply", [_i(kind), _i(inverse)] + _fl(m, 16) + _fl(m_inv, 16) + _fl(v, 3), 3)[1])

    def orc_fn_sample_discrete(self, weights, n, u, pmf, u_remapped):
        r, o = self._run("sample_discrete", [_i(n), _f(u)] + (_fl(weights, n) if n > 0 else [0]), 2)
        f = o.view(np.float32)
        for dst, val in ((pmf, f[0]), (u_remapped, f[1])):
            if dst is not None:
                _deref(dst).value = float(val)
        return r if r < 0x80000000 else r - (1 << 32)

    def orc_fn_sampler_stream(self, px, py, sample_index, seed, n, out):
        o = self._run("sampler_stream", [_i(px), _i(py), _i(sample_index), int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF, _i(n)], max(1, n))[1]
        self._store(out, o, n)
        return float(o.view(np.float32)[0]) if n > 0 else 0.0

    def _unary(self, which, x): return self._scalar("unary", [which, _f(x)])
    def orc_fn_sin(self, x): return self._unary(0, x)
    def orc_fn_cos(self, x): return self._unary(1, x)
    def orc_fn_asin(self, x): return self._unary(2, x)
    def orc_fn_acos(self, x): return self._unary(3, x)
    def orc_fn_exp(self, x): return self._unary(4, x)
    def orc_fn_log(self, x): return self._unary(5, x)
    def orc_fn_atanh(self, x): return self._unary(6, x)
    def orc_fn_cosh(self, x): return self._unary(7, x)
    def orc_fn_log2(self, x): return self._unary(8, x)

    # ---- traversal leaves ----
    def orc_fn_intersect_p_cached(self, bmin, bmax, o, d, t_max):
        return self._run("intersect_p_cached", _fl(bmin, 3) + _fl(bmax, 3) + _fl(o, 3) + _fl(d, 3) + [_f(t_max)], 1)[0]

    def orc_fn_intersect_triangle(self, o, d, t_max, p0, p1, p2, out4):
        r, w = self._run("intersect_triangle", _fl(o, 3) + _fl(d, 3) + [_f(t_max)] + _fl(p0, 3) + _fl(p1, 3) + _fl(p2, 3), 4)
        if r:
            self._store(out4, w)
        return r

    def orc_fn_blp_intersect(self, pts, o, d, t_max, out3):
        r, w = self._run("blp_intersect", _fl(pts, 12) + _fl(o, 3) + _fl(d, 3) + [_f(t_max)], 3)
        if r:
            self._store(out3, w)
        return r

    # ---- scattering.rs / bxdf.rs ----
    def orc_fn_tr_d(self, ax, ay, wm): return self._scalar("tr_d", [_f(ax), _f(ay)] + _fl(wm, 3))
    def orc_fn_tr_g(self, ax, ay, wo, wi): return self._scalar("tr_g", [_f(ax), _f(ay)] + _fl(wo, 3) + _fl(wi, 3))
    def orc_fn_tr_lambda(self, ax, ay, w): return self._scalar("tr_lambda", [_f(ax), _f(ay)] + _fl(w, 3))
    def orc_fn_fresnel_dielectric(self, c, eta): return self._scalar("fresnel_dielectric", [_f(c), _f(eta)])
    def orc_fn_fresnel_complex(self, c, eta, k): return self._scalar("fresnel_complex", [_f(c), _f(eta), _f(k)])

    def orc_fn_tr_sample_wm(self, ax, ay, w, u, out):
        self._store(out, self._run("tr_sample_wm", [_f(ax), _f(ay)] + _fl(w, 3) + _fl(u, 2), 3)[1])

    @staticmethod
    def _bxdf(kind, r4, k4, eta, ax, ay):
        return [_i(kind)] + _fl(r4, 4) + _fl(k4, 4) + [_f(eta), _f(ax), _f(ay)]

    def orc_fn_bxdf_sample_f(self, kind, r4, k4, eta, ax, ay, wo, uc, u, out10):
        r, w = self._run("bxdf_sample_f", self._bxdf(kind, r4, k4, eta, ax, ay) + _fl(wo, 3) + [_f(uc)] + _fl(u, 2), 10)
        if r:
            self._store(out10, w)
        return r

    def orc_fn_bxdf_f_pdf(self, kind, r4, k4, eta, ax, ay, wo, wi, out5):
        self._store(out5, self._run("bxdf_f_pdf", self._bxdf(kind, r4, k4, eta, ax, ay) + _fl(wo, 3) + _fl(wi, 3), 5)[1])

    @staticmethod
    def _layered(kind, p, ip):
        return [_i(kind)] + _fl(p, 19) + [_i(ip[0]), _i(ip[1])]

    def orc_fn_layered_f_pdf(self, kind, p, ip, wo, wi, out6):
        self._store(out6, self._run("layered_f_pdf", self._layered(kind, p, ip) + _fl(wo, 3) + _fl(wi, 3), 6)[1])

    def orc_fn_layered_sample_f(self, kind, p, ip, wo, uc, u, out10):
        r, w = self._run("layered_sample_f", self._layered(kind, p, ip) + _fl(wo, 3) + [_f(uc)] + _fl(u, 2), 10)
        if r:
            self._store(out10, w)
        return r

    # ---- shapes / lights / camera / film ----
    def orc_fn_offset_ray_origin(self, p, err, n, w, out3):
        self._store(out3, self._run("offset_ray_origin", _fl(p, 3) + _fl(err, 3) + _fl(n, 3) + _fl(w, 3), 3)[1])

    def orc_fn_triangle_sample_with_context(self, p0, p1, p2, ctx_p, ctx_n, ctx_ns, u, out7):
        r, w = self._run("triangle_sample_with_context", _fl(p0, 3) + _fl(p1, 3) + _fl(p2, 3) + _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(u, 2), 7)
        if r:
            self._store(out7, w)
        return r

    def orc_fn_triangle_pdf_with_context(self, p0, p1, p2, ctx_p, ctx_n, ctx_ns, wi):
        return self._scalar("triangle_pdf_with_context", _fl(p0, 3) + _fl(p1, 3) + _fl(p2, 3) + _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(wi, 3))

    def orc_fn_triangle_interaction(self, p9, n9, s9, uv6, flip, b3, wo, out):
        words = _fl(p9, 9) + [1 if n9 is not None else 0] + _fl(n9, 9) + [1 if s9 is not None else 0] + _fl(s9, 9) + [1 if uv6 is not None else 0] + _fl(uv6, 6) + \
                [_i(flip)] + _fl(b3, 3) + _fl(wo, 3)
        self._store(out, self._run("triangle_interaction", words, 38)[1])

    def orc_fn_sphere_sample_with_context(self, sp, ctx_p, ctx_n, ctx_ns, u, out7):
        r, w = self._run("sphere_sample_with_context", _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(u, 2) + _struct_words(_deref(sp)), 7)
        if r:
            self._store(out7, w)
        return r

    def orc_fn_sphere_pdf_with_context(self, sp, ctx_p, ctx_n, ctx_ns, wi):
        return self._scalar("sphere_pdf_with_context", _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(wi, 3) + _struct_words(_deref(sp)))

    def orc_fn_blp_sample_with_context(self, pts, flip, ctx_p, ctx_n, ctx_ns, u, out7):
        r, w = self._run("blp_sample_with_context", _fl(pts, 12) + [_i(flip)] + _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(u, 2), 7)
        if r:
            self._store(out7, w)
        return r

    def orc_fn_blp_pdf_with_context(self, pts, flip, ctx_p, ctx_n, ctx_ns, wi):
        return self._scalar("blp_pdf_with_context", _fl(pts, 12) + [_i(flip)] + _fl(ctx_p, 3) + _fl(ctx_n, 3) + _fl(ctx_ns, 3) + _fl(wi, 3))

    def orc_fn_area_light_l(self, two_sided, scale, table, n_table, lambda_min, n, w, lambda4, out4):
        words = [_i(two_sided), _f(scale), _i(n_table), _i(lambda_min)] + _fl(n, 3) + _fl(w, 3) + _fl(lambda4, 4) + _fl(table, n_table)
        self._store(out4, self._run("area_light_l", words, 4)[1])

    def orc_fn_film_add_sample(self, r_bar, g_bar, b_bar, imaging_ratio, max_component_value, L4, lambda4, pdf4, weight, pixel4, rgb_out3):
        px = struct.unpack("<8I", struct.pack("<4d", *[pixel4[k] for k in range(4)]))
        words = [_f(imaging_ratio), _f(max_component_value)] + _fl(L4, 4) + _fl(lambda4, 4) + _fl(pdf4, 4) + [_f(weight)] + list(px) + \
                _fl(r_bar, 471) + _fl(g_bar, 471) + _fl(b_bar, 471)
        o = self._run("film_add_sample", words, 11)[1]
        new = struct.unpack("<4d", o[:8].tobytes())
        for k in range(4):
            pixel4[k] = new[k]
        if rgb_out3 is not None:
            self._store(rgb_out3, o[8:11])

    def orc_fn_camera_ray_differential(self, cam, p_film, p_lens, out18):
        self._store(out18, self._run("camera_ray_differential", _fl(p_film, 2) + _fl(p_lens, 2) + _struct_words(_deref(cam)), 18)[1])
