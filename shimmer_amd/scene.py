"""Host-side scene assembly: the marshalling a Rust `impl Integrator` would do before calling the C ABI.

Mirrors (paths relative to /root/reference/src) the pieces of scene construction that sit directly before
the hot path: primitive bounds (shape/triangle.rs:507-510, shape/sphere.rs:275-280), BvhAggregate::new via
shm_bvh_build (aggregate.rs:207-467), one DiffuseAreaLight per emissive shape (loading/scene.rs:609-624),
DenselySampledSpectrum::new (spectra/spectrum.rs:179-196), BlackbodySpectrum (:430-489),
PiecewiseLinearSpectrum::from_interleaved (:313-352), spectrum_to_photometric (:617-631).
Everything computed here is INPUT data handed identically to the HIP library and to the CPU oracle.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from . import abi

_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = dict(np.load(Path(__file__).resolve().parent / "data" / "spectral_tables.npz"))
    return _TABLES


f32 = np.float32


def blackbody_dense(temperature):
    """DenselySampledSpectrum::new(BlackbodySpectrum::new(T)) at 360..830 nm, f32 as the reference computes it (spectra/spectrum.rs:430-489),
    through the host library (shm_blackbody_dense): the one implementation the C++ PBRT loader uses too, so that a generated scene and the
    same scene loaded from .pbrt text carry bit-identical emission tables."""
    out = np.zeros(471, np.float32)
    lib = abi.load_library()
    abi.check(lib, lib.shm_blackbody_dense(float(temperature), out.ctypes.data_as(abi.c_float_p)), "shm_blackbody_dense")
    return out


def piecewise_from_interleaved(samples, normalize=False):
    """PiecewiseLinearSpectrum::from_interleaved -> (lambdas, values)."""
    s = np.asarray(samples, dtype=np.float32)
    lam, val = list(s[0::2]), list(s[1::2])
    if lam[0] > 360.0:
        lam.insert(0, f32(359.0))
        val.insert(0, val[0])
    if lam[-1] < 830.0:
        lam.append(f32(831.0))
        val.append(val[-1])
    lam, val = np.asarray(lam, np.float32), np.asarray(val, np.float32)
    if normalize:
        dense = piecewise_to_dense(lam, val)
        y = tables()["CIE_Y"]
        integral = f32(0.0)
        for a, b in zip(dense, y):
            integral = f32(integral + f32(a * b))
        val = (val * f32(tables()["CIE_Y_INTEGRAL"] / integral)).astype(np.float32)
    return lam, val


def piecewise_get(lam, val, x):
    """PiecewiseLinearSpectrum::get for scalar x (f32)."""
    x = f32(x)
    if x < lam[0] or x > lam[-1]:
        return f32(0.0)
    o = int(np.clip(np.searchsorted(lam, x, side="right") - 1, 0, len(lam) - 2))
    t = f32((x - lam[o]) / (lam[o + 1] - lam[o]))
    return f32(val[o] * f32(f32(1.0) - t) + f32(val[o + 1] * t))


def piecewise_to_dense(lam, val):
    return np.asarray([piecewise_get(lam, val, float(l)) for l in range(360, 831)], dtype=np.float32)


def spectrum_to_photometric(dense):
    """spectrum_to_photometric over a 360..830 dense table (f32 running sum, as the reference)."""
    y = tables()["CIE_Y"]
    acc = f32(0.0)
    for a, b in zip(y, dense):
        acc = f32(acc + f32(a * b))
    return acc


def generate_pyramid(image):
    """Image::generate_pyramid (image.rs:699-800) for a float image whose sides are powers of two (other sizes go through
    float_resize_up first in the reference: host-side, not mirrored): each level is the 2x2 box average of the previous one,
    f32, summed in the reference's order; the last level is 1x1."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[:2]
    assert h & (h - 1) == 0 and w & (w - 1) == 0, "power-of-two sides only"
    levels = [img]
    while img.shape[0] > 1 or img.shape[1] > 1:
        h, w = img.shape[:2]
        nh, nw = max(1, (h + 1) // 2), max(1, (w + 1) // 2)
        y0, x0 = 2 * np.arange(nh), 2 * np.arange(nw)
        y1, x1 = (y0 + 1 if h > 1 else y0), (x0 + 1 if w > 1 else x0)
        a, b, c, dd = img[y0][:, x0], img[y0][:, x1], img[y1][:, x0], img[y1][:, x1]
        img = (f32(0.25) * (((a + b).astype(np.float32) + c).astype(np.float32) + dd).astype(np.float32)).astype(np.float32)
        levels.append(img)
    return levels


def _as_f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _fptr(a):
    return a.ctypes.data_as(abi.c_float_p)


IDENTITY = np.eye(4, dtype=np.float32)


class SceneBuilder:
    """Accumulates shapes/materials/lights in input order, builds the BVH with the host-side mirror of
    BvhAggregate::new and emits a ShmSceneDesc (keeping every backing array alive)."""

    def __init__(self):
        self.meshes = []       # dict(p, vi, n, s, uv, reverse, swaps)
        self.spheres = []      # abi.ShmSphere
        self.patch_meshes = [] # dict(p, vi (N,4), reverse, swaps)
        self._patch_count = 0
        self.prims = []        # chunks of (N,4) int64: shape_kind, shape_index, material, area_light
        self._n_prims = 0
        self.materials = []    # abi.ShmMaterial
        self.lights = []       # abi.ShmLight (primitive = INPUT prim index until build())
        self.spec = []         # list of float32 arrays (spectrum pool)
        self.spec_len = 0
        self.camera = None
        self.film = None
        self._keep = []
        self._tri_count = 0
        self._emission_cache = {}
        self.textures = []      # abi.ShmImageTexture
        self.tex_levels = []    # (width, height, offset)
        self.texels = []        # float32 arrays
        self.texel_len = 0
        self.color_space = None  # dict(res, scale, data, illuminant)
        self.image_lights = []  # abi.ShmImageInfiniteLight
        self.float_textures = []  # abi.ShmFloatTexture
        self.spectrum_textures = []  # abi.ShmSpectrumTexture
        self.owners = []        # per chunk of self.prims: 0 = the scene, k = object definition k (ObjectBegin / ObjectEnd)
        self.objects = {}       # name -> k
        self._object = 0
        self.instances = []     # (object id, render_from_instance 4x4 f32)

    # ---- spectra ----
    def spectrum_constant(self, c):
        s = abi.ShmSpectrum()
        s.kind, s.c = abi.SHM_SPECTRUM_CONSTANT, float(c)
        return s

    def _pool(self, arr):
        off = self.spec_len
        arr = _as_f32(arr).ravel()
        self.spec.append(arr)
        self.spec_len += arr.size
        return off

    def spectrum_dense(self, values471):
        v = _as_f32(values471)
        assert v.size == 471
        s = abi.ShmSpectrum()
        s.kind, s.offset, s.n, s.lambda_min = abi.SHM_SPECTRUM_DENSE, self._pool(v), 471, 360
        return s

    def spectrum_piecewise(self, lam, val):
        lam, val = _as_f32(lam), _as_f32(val)
        assert lam.size == val.size and lam.size >= 2
        s = abi.ShmSpectrum()
        s.kind, s.offset, s.n = abi.SHM_SPECTRUM_PIECEWISE_LINEAR, self._pool(np.concatenate([lam, val])), lam.size
        return s

    def spectrum_rgb(self, coeffs, scale=None, illuminant=None):
        """RGB-derived spectra from sigmoid coefficients (c0, c1, c2) the host looked up with RgbColorSpace::to_rgb_coeffs:
        RgbAlbedoSpectrum (no scale), RgbUnboundedSpectrum (scale), RgbIlluminantSpectrum (scale + dense illuminant table)."""
        s = abi.ShmSpectrum()
        s.rgb_c[:] = [float(f32(c)) for c in coeffs]
        if illuminant is not None:
            d = self.spectrum_dense(illuminant)
            s.kind, s.c, s.offset, s.n, s.lambda_min = abi.SHM_SPECTRUM_RGB_ILLUMINANT, float(scale), d.offset, d.n, d.lambda_min
        elif scale is not None:
            s.kind, s.c = abi.SHM_SPECTRUM_RGB_UNBOUNDED, float(scale)
        else:
            s.kind = abi.SHM_SPECTRUM_RGB_ALBEDO
        return s

    # ---- image textures (ABI v6) ----
    def use_srgb_color_space(self):
        """RgbColorSpace::SRGB as the device needs it: the rgb2spec coefficient table at the reference's resolution 64 (tools/gen_rgb2spec.py;
        the reference loads rgbtospec/srgb.spec, rgb_to_spectra.rs:27-31) and the D65 illuminant, densely sampled."""
        if self.color_space is None:
            f = Path(__file__).resolve().parent / "data" / "rgb2spec_srgb_res64.npz"
            if not f.exists():  # generated once by __graft_entry__.build() (tools/gen_rgb2spec.py, about two minutes on three cores)
                import subprocess, sys
                subprocess.check_call([sys.executable, str(Path(__file__).resolve().parents[1] / "tools" / "gen_rgb2spec.py"), "64"])
            t = np.load(f)
            # RgbColorSpace::SRGB.illuminant = DenselySampledSpectrum::new(StdIllum-D65), the normalised table (colorspace.rs:65, 139;
            # named_spectrum.rs:52-56) — the same floats the C++ front end hands over (host/scene_assembly.hpp illuminant_d65_dense)
            illum = piecewise_to_dense(*piecewise_from_interleaved(tables()["CIE_ILLUM_D6500"], True))
            self.color_space = dict(res=int(t["res"]), scale=_as_f32(t["scale"]), data=_as_f32(t["data"]).ravel(), illuminant=illum)
        return self.color_space

    def add_image_texture(self, image, filter="bilinear", wrap="repeat", scale=1.0, invert=False, spectrum_type="albedo",
                          mapping="uv", su=1.0, sv=1.0, du=0.0, dv=0.0, max_anisotropy=8.0, texture_from_render=None,
                          vs=(1.0, 0.0, 0.0), vt=(0.0, 1.0, 0.0), color_space=True, pyramid=None):
        """SpectrumImageTexture::create (texture.rs:728-775) with its defaults. `image`: (H, W) or (H, W, 3) linear float values,
        row 0 = top. Returns the ShmSpectrum that binds the texture to a material's spectrum slot. `pyramid` overrides the levels
        (list of arrays, finest first, ending in 1x1)."""
        levels = pyramid if pyramid is not None else generate_pyramid(image)
        nc = 1 if levels[0].ndim == 2 else levels[0].shape[2]
        assert nc in (1, 3)
        t = abi.ShmImageTexture()
        t.mapping = {"uv": abi.SHM_TEXMAP_UV, "spherical": abi.SHM_TEXMAP_SPHERICAL, "cylindrical": abi.SHM_TEXMAP_CYLINDRICAL,
                     "planar": abi.SHM_TEXMAP_PLANAR}[mapping]
        t.su, t.sv, t.du, t.dv = float(su), float(sv), float(du), float(dv)
        t.vs[:], t.vt[:] = [float(x) for x in vs], [float(x) for x in vt]
        t.texture_from_render[:] = [float(x) for x in _as_f32(IDENTITY if texture_from_render is None else texture_from_render).ravel()]
        t.filter = {"point": abi.SHM_TEXFILTER_POINT, "bilinear": abi.SHM_TEXFILTER_BILINEAR, "trilinear": abi.SHM_TEXFILTER_TRILINEAR,
                    "ewa": abi.SHM_TEXFILTER_EWA}[filter.lower()]
        t.max_anisotropy = float(max_anisotropy)
        t.wrap = {"black": abi.SHM_WRAP_BLACK, "clamp": abi.SHM_WRAP_CLAMP, "repeat": abi.SHM_WRAP_REPEAT,
                  "octahedralsphere": abi.SHM_WRAP_OCTAHEDRAL_SPHERE}[wrap]
        t.scale, t.invert = float(scale), int(bool(invert))
        t.spectrum_type = {"albedo": abi.SHM_SPECTRUM_TYPE_ALBEDO, "unbounded": abi.SHM_SPECTRUM_TYPE_UNBOUNDED,
                           "illuminant": abi.SHM_SPECTRUM_TYPE_ILLUMINANT}[spectrum_type]
        t.n_channels = nc
        t.has_color_space = int(bool(color_space))
        if color_space:
            self.use_srgb_color_space()
        t.first_level, t.n_levels = len(self.tex_levels), len(levels)
        for lv in levels:
            a = _as_f32(lv)
            self.tex_levels.append((a.shape[1], a.shape[0], self.texel_len))
            self.texels.append(a.ravel())
            self.texel_len += a.size
        self.textures.append(t)
        sp = abi.ShmSpectrum()
        sp.kind, sp.offset = abi.SHM_SPECTRUM_IMAGE_TEXTURE, len(self.textures) - 1
        return sp

    def spectrum_named(self, name):
        """NamedSpectrum (spectra/named_spectrum.rs:13-27): metals / glasses as PiecewiseLinear."""
        key = {"glass-BK7": "GLASS_BK7_ETA_SAMPLES", "glass-BAF10": "GLASS_BAF10_ETA_SAMPLES", "glass-F11": "GLASS_F11_ETA_SAMPLES",
               "metal-Cu-eta": "CU_ETA_SAMPLES", "metal-Cu-k": "CU_K_SAMPLES", "metal-Au-eta": "AU_ETA_SAMPLES",
               "metal-Au-k": "AU_K_SAMPLES", "metal-Ag-eta": "AG_ETA_SAMPLES", "metal-Ag-k": "AG_K_SAMPLES",
               "metal-Al-eta": "AL_ETA_SAMPLES", "metal-Al-k": "AL_K_SAMPLES"}[name]
        lam, val = piecewise_from_interleaved(tables()[key], False)
        return self.spectrum_piecewise(lam, val)

    # ---- materials ----
    def material_diffuse(self, reflectance):
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_DIFFUSE
        m.has_displacement, m.displacement = 1, 0.0  # material.rs:280 always installs a constant-0 displacement
        m.a = reflectance if isinstance(reflectance, abi.ShmSpectrum) else self.spectrum_constant(reflectance)
        self.materials.append(m)
        return len(self.materials) - 1

    def material_conductor(self, eta, k, roughness=0.0, remap=True):
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_CONDUCTOR
        m.remap_roughness, m.u_roughness, m.v_roughness = int(remap), float(roughness), float(roughness)
        m.a, m.b = eta, k
        self.materials.append(m)
        return len(self.materials) - 1

    def material_dielectric(self, eta, roughness=0.0, remap=True, thin=False):
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_THIN_DIELECTRIC if thin else abi.SHM_MATERIAL_DIELECTRIC
        m.remap_roughness, m.u_roughness, m.v_roughness = int(remap), float(roughness), float(roughness)
        m.a = eta if isinstance(eta, abi.ShmSpectrum) else self.spectrum_constant(eta)
        self.materials.append(m)
        return len(self.materials) - 1

    def _spec(self, v):
        return v if isinstance(v, abi.ShmSpectrum) else self.spectrum_constant(v)

    def material_coated_diffuse(self, reflectance=0.5, roughness=0.0, thickness=0.01, eta=1.5, g=0.0, albedo=0.0, max_depth=10,
                                n_samples=1, remap=True):
        """CoatedDiffuseMaterial::create defaults (material.rs:826-911): no displacement texture unless given."""
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_COATED_DIFFUSE
        m.remap_roughness, m.u_roughness, m.v_roughness = int(remap), float(roughness), float(roughness)
        m.thickness, m.g, m.max_depth, m.n_samples = float(thickness), float(g), int(max_depth), int(n_samples)
        m.a, m.c, m.d = self._spec(reflectance), self._spec(albedo), self._spec(eta)
        m.b = self.spectrum_constant(0.0)
        self.materials.append(m)
        return len(self.materials) - 1

    def material_coated_conductor(self, conductor_eta=None, k=None, reflectance=None, interface_roughness=0.0, conductor_roughness=0.0,
                                  thickness=0.01, interface_eta=1.5, g=0.0, albedo=0.0, max_depth=10, n_samples=1, remap=True):
        """CoatedConductorMaterial::create (material.rs:1052-1180): conductor given by (eta, k) spectra — default
        metal-Cu — or by a reflectance."""
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_COATED_CONDUCTOR
        m.remap_roughness, m.u_roughness, m.v_roughness = int(remap), float(interface_roughness), float(interface_roughness)
        m.u2_roughness, m.v2_roughness = float(conductor_roughness), float(conductor_roughness)
        m.thickness, m.g, m.max_depth, m.n_samples = float(thickness), float(g), int(max_depth), int(n_samples)
        if reflectance is not None:
            m.conductor_from_reflectance = 1
            m.a, m.b = self._spec(reflectance), self.spectrum_constant(0.0)
        else:
            m.a = conductor_eta if conductor_eta is not None else self.spectrum_named("metal-Cu-eta")
            m.b = k if k is not None else self.spectrum_named("metal-Cu-k")
        m.c, m.d = self._spec(albedo), self._spec(interface_eta)
        self.materials.append(m)
        return len(self.materials) - 1

    def material_mix(self, m0, m1, amount=0.5):
        """MixMaterial::create (material.rs:1296-1306): two material indices (either may be a mix) and a constant amount."""
        m = abi.ShmMaterial()
        m.kind = abi.SHM_MATERIAL_MIX
        m.mix_material[0], m.mix_material[1], m.mix_amount = int(m0), int(m1), float(amount)
        z = self.spectrum_constant(0.0)
        m.a, m.b, m.c, m.d = z, z, z, z
        self.materials.append(m)
        return len(self.materials) - 1

    # ---- lights ----
    def _area_light(self, prim_index, area, dense_emission, scale, two_sided):
        """DiffuseAreaLight::create: scale /= spectrum_to_photometric(L) (light.rs:598)."""
        l = abi.ShmLight()
        l.kind, l.primitive, l.two_sided, l.area = abi.SHM_LIGHT_DIFFUSE_AREA, prim_index, int(two_sided), float(area)
        key = id(dense_emission)
        if key not in self._emission_cache:  # one pooled table + photometric integral per distinct emission spectrum
            self._emission_cache[key] = (self.spectrum_dense(dense_emission), spectrum_to_photometric(dense_emission), dense_emission)
        spec, photometric, _ = self._emission_cache[key]
        l.scale = float(f32(scale) / photometric)
        l.spectrum = spec
        self.lights.append(l)
        return len(self.lights) - 1

    def light_point(self, position, dense_intensity, scale=1.0):
        l = abi.ShmLight()
        l.kind = abi.SHM_LIGHT_POINT
        l.scale = float(f32(scale) / spectrum_to_photometric(dense_intensity))
        l.position[:] = [float(x) for x in position]
        l.spectrum = self.spectrum_dense(dense_intensity)
        self.lights.append(l)
        return len(self.lights) - 1

    def light_uniform_infinite(self, dense_emission, scale=1.0):
        l = abi.ShmLight()
        l.kind = abi.SHM_LIGHT_UNIFORM_INFINITE
        l.scale = float(f32(scale) / spectrum_to_photometric(dense_emission))
        l.spectrum = self.spectrum_dense(dense_emission)
        self.lights.append(l)
        return len(self.lights) - 1

    # ---- float textures (texture.rs:88-305) ----
    def _ftex(self, kind, value=0.0, a=0, b=0, c=0, dir=(0.0, 1.0, 0.0), image=0):
        t = abi.ShmFloatTexture()
        t.kind, t.value, t.a, t.b, t.c, t.image = kind, float(value), int(a), int(b), int(c), int(image)
        t.dir[:] = [float(x) for x in dir]
        self.float_textures.append(t)
        return len(self.float_textures) - 1

    def _ftex_of(self, v):
        return v if isinstance(v, (int, np.integer)) and not isinstance(v, bool) else self.ftex_constant(v)

    def ftex_constant(self, value):
        return self._ftex(abi.SHM_FLOATTEX_CONSTANT, value=value)

    def ftex_scaled(self, tex=1.0, scale=1.0):
        """"scale": both parameters are float textures (handles, i.e. ints) or numbers (floats)."""
        a, b = self._ftex_of(tex), self._ftex_of(scale)
        return self._ftex(abi.SHM_FLOATTEX_SCALED, a=a, b=b)

    def ftex_mix(self, tex1=0.0, tex2=1.0, amount=0.5):
        a, b, c = self._ftex_of(tex1), self._ftex_of(tex2), self._ftex_of(amount)
        return self._ftex(abi.SHM_FLOATTEX_MIX, a=a, b=b, c=c)

    def ftex_direction_mix(self, tex1=0.0, tex2=1.0, dir=(0.0, 1.0, 0.0)):
        a, b = self._ftex_of(tex1), self._ftex_of(tex2)
        return self._ftex(abi.SHM_FLOATTEX_DIRECTION_MIX, a=a, b=b, dir=dir)

    def ftex_image(self, image, **kw):
        """"imagemap" float texture: the same options as add_image_texture (FloatImageTexture::create, texture.rs:345-391)."""
        kw.setdefault("color_space", False)
        sp = self.add_image_texture(image, **kw)
        return self._ftex(abi.SHM_FLOATTEX_IMAGE, image=sp.offset)

    # ---- composite spectrum textures (texture.rs:536-687) ----
    def _stex_node(self, v):
        """A spectrum-texture operand -> node index: a bound ShmSpectrum of kind TEXTURE_NODE is its node, anything else becomes a leaf."""
        sp = self._spec(v)
        if sp.kind == abi.SHM_SPECTRUM_TEXTURE_NODE:
            return sp.offset
        t = abi.ShmSpectrumTexture()
        t.kind, t.leaf = abi.SHM_SPECTEX_LEAF, sp
        self.spectrum_textures.append(t)
        return len(self.spectrum_textures) - 1

    def _stex(self, kind, a, b=0, f=0, dir=(0.0, 1.0, 0.0)):
        t = abi.ShmSpectrumTexture()
        t.kind, t.a, t.b, t.f = kind, int(a), int(b), int(f)
        t.dir[:] = [float(x) for x in dir]
        self.spectrum_textures.append(t)
        sp = abi.ShmSpectrum()
        sp.kind, sp.offset = abi.SHM_SPECTRUM_TEXTURE_NODE, len(self.spectrum_textures) - 1
        return sp

    def stex_scaled(self, tex, scale=1.0):
        return self._stex(abi.SHM_SPECTEX_SCALED, self._stex_node(tex), f=self._ftex_of(scale))

    def stex_mix(self, tex1, tex2, amount=0.5):
        return self._stex(abi.SHM_SPECTEX_MIX, self._stex_node(tex1), self._stex_node(tex2), f=self._ftex_of(amount))

    def stex_direction_mix(self, tex1, tex2, dir=(0.0, 1.0, 0.0)):
        return self._stex(abi.SHM_SPECTEX_DIRECTION_MIX, self._stex_node(tex1), self._stex_node(tex2), dir=dir)

    def set_float_texture(self, material, slot, ftex):
        """Bind a float texture handle to one of a material's float parameters (abi.SHM_FLOATSLOT_*)."""
        m = self.materials[material]
        m.float_tex[slot] = int(ftex) + 1
        if slot == abi.SHM_FLOATSLOT_DISPLACEMENT:
            m.has_displacement = 1

    def set_normal_map(self, material, image):
        """"normalmap": an RGB image; only reached when the material has no displacement (interaction.rs:223-236)."""
        sp = self.add_image_texture(image, color_space=False)
        self.materials[material].normal_map = sp.offset + 1

    def light_image_infinite(self, image, scale=1.0, render_from_light=None):
        """ImageInfinitelight (light.rs:805-981; created by the "infinite" light with a filename, light.rs:147-190): `image` is a
        square (n, n, 3) linear-RGB environment map in the equal-area octahedral layout, row 0 first; `scale` is the final scale."""
        img = _as_f32(image)
        assert img.ndim == 3 and img.shape[0] == img.shape[1] and img.shape[2] == 3
        self.use_srgb_color_space()
        il = abi.ShmImageInfiniteLight()
        m = np.asarray(IDENTITY if render_from_light is None else render_from_light, np.float64).reshape(4, 4)
        il.render_from_light[:] = [float(x) for x in _as_f32(m).ravel()]
        il.light_from_render[:] = [float(x) for x in _as_f32(np.linalg.inv(m)).ravel()]
        il.image_level = len(self.tex_levels)
        self.tex_levels.append((img.shape[1], img.shape[0], self.texel_len))
        self.texels.append(img.ravel())
        self.texel_len += img.size
        self.image_lights.append(il)
        l = abi.ShmLight()
        l.kind, l.primitive, l.scale = abi.SHM_LIGHT_IMAGE_INFINITE, len(self.image_lights) - 1, float(scale)
        self.lights.append(l)
        return len(self.lights) - 1

    # ---- shapes ----
    def add_mesh(self, p, vi, material, n=None, s=None, uv=None, reverse_orientation=False, swaps_handedness=False,
                 emission=None, emission_scale=1.0, two_sided=False):
        """trianglemesh: vertices already in render space. emission = dense 471 table -> one area light per triangle."""
        p = _as_f32(p, (-1, 3))
        vi = np.ascontiguousarray(vi, dtype=np.uint32).reshape(-1, 3)
        mesh = dict(p=p, vi=vi, n=None if n is None else _as_f32(n, (-1, 3)), s=None if s is None else _as_f32(s, (-1, 3)),
                    uv=None if uv is None else _as_f32(uv, (-1, 2)), reverse=bool(reverse_orientation), swaps=bool(swaps_handedness))
        self.meshes.append(mesh)
        base = self._tri_count
        ntri = vi.shape[0]
        self._tri_count += ntri
        first_prim = self._n_prims
        chunk = np.empty((ntri, 4), np.int64)
        chunk[:, 0], chunk[:, 1], chunk[:, 2], chunk[:, 3] = abi.SHM_SHAPE_TRIANGLE, base + np.arange(ntri), material, -1
        if emission is not None:
            p0, p1, p2 = p[vi[:, 0]], p[vi[:, 1]], p[vi[:, 2]]
            for t in range(ntri):
                # Triangle::area = 0.5 * |(p1-p0) x (p2-p0)| — only the light's `area` field (phi(); unused by the path)
                area = 0.5 * float(np.linalg.norm(np.cross((p1[t] - p0[t]).astype(np.float64), (p2[t] - p0[t]).astype(np.float64))))
                chunk[t, 3] = self._area_light(first_prim + t, area, emission, emission_scale, two_sided)
        self.prims.append(chunk)
        self.owners.append(self._object)
        self._n_prims += ntri
        return first_prim

    def add_ply(self, lib, filename, material, render_from_object=None, reverse_orientation=False, **light):
        """The "plymesh" shape (shape/shape.rs:97-135): TriQuadMesh::read_ply through the host mirror, then a TriangleMesh of its
        triangles and a BilinearPatchMesh of its quads, both with the file's n and uv (zeros where the file has none: the reference
        hands those over too). Positions go to render space as TriangleMesh::new does (mesh.rs:42-64): points through the matrix,
        normals through the inverse transpose, negated under reverse_orientation. Returns (first triangle prim, first patch prim)."""
        m = abi.ShmPlyMesh()
        abi.check(lib, lib.shm_ply_read(str(filename).encode(), C.byref(m)), "shm_ply_read")
        try:
            nv = m.n_vertices
            p = np.ctypeslib.as_array(m.p, shape=(nv, 3)).copy() if nv else np.zeros((0, 3), np.float32)
            n = np.ctypeslib.as_array(m.n, shape=(nv, 3)).copy() if nv else np.zeros((0, 3), np.float32)
            uv = np.ctypeslib.as_array(m.uv, shape=(nv, 2)).copy() if nv else np.zeros((0, 2), np.float32)
            tri = np.ctypeslib.as_array(m.tri_indices, shape=(m.n_tri_indices,)).copy().reshape(-1, 3) if m.n_tri_indices else None
            quad = np.ctypeslib.as_array(m.quad_indices, shape=(m.n_quad_indices,)).copy().reshape(-1, 4) if m.n_quad_indices else None
        finally:
            lib.shm_ply_free(C.byref(m))
        rfo = _as_f32(IDENTITY if render_from_object is None else render_from_object, (4, 4))
        inv = _as_f32(np.linalg.inv(rfo.astype(np.float64)), (4, 4))
        # apply_point_helper / apply_normal_helper in f32, left to right as written (transform.rs:753-786)
        pr = ((rfo[None, :3, 0] * p[:, 0:1] + rfo[None, :3, 1] * p[:, 1:2]).astype(np.float32) + rfo[None, :3, 2] * p[:, 2:3]).astype(np.float32) + rfo[None, :3, 3]
        nr = ((inv[None, 0, :3] * n[:, 0:1] + inv[None, 1, :3] * n[:, 1:2]).astype(np.float32) + inv[None, 2, :3] * n[:, 2:3]).astype(np.float32)
        if reverse_orientation:
            nr = -nr
        swaps = bool(np.linalg.det(rfo[:3, :3].astype(np.float64)) < 0)
        first_tri = first_patch = None
        if tri is not None:
            first_tri = self.add_mesh(pr.astype(np.float32), tri.astype(np.uint32), material, n=nr, uv=uv, reverse_orientation=reverse_orientation,
                                      swaps_handedness=swaps, **light)
        if quad is not None:
            first_patch = self.add_patch_mesh(pr.astype(np.float32), quad.astype(np.uint32), material, n=nr, uv=uv,
                                              reverse_orientation=reverse_orientation, swaps_handedness=swaps, **light)
        return first_tri, first_patch

    def add_patch_mesh(self, p, vi, material, n=None, uv=None, reverse_orientation=False, swaps_handedness=False, emission=None,
                       emission_scale=1.0, two_sided=False):
        """bilinearmesh (shape/mesh.rs:289-376): vertices in render space, 4 indices per patch in the order p00, p10, p01, p11
        (bilinear_patch.rs:87-98). emission -> one DiffuseAreaLight per patch."""
        p = _as_f32(p, (-1, 3))
        vi = np.ascontiguousarray(vi, dtype=np.uint32).reshape(-1, 4)
        self.patch_meshes.append(dict(p=p, vi=vi, n=None if n is None else _as_f32(n, (-1, 3)), uv=None if uv is None else _as_f32(uv, (-1, 2)),
                                      reverse=bool(reverse_orientation), swaps=bool(swaps_handedness)))
        base, n = self._patch_count, vi.shape[0]
        self._patch_count += n
        first_prim = self._n_prims
        chunk = np.empty((n, 4), np.int64)
        chunk[:, 0], chunk[:, 1], chunk[:, 2], chunk[:, 3] = abi.SHM_SHAPE_BILINEAR_PATCH, base + np.arange(n), material, -1
        if emission is not None:
            for t in range(n):
                q = p[vi[t]].astype(np.float64)  # the light's `area` field feeds phi() only (unused by the path): float64 estimate
                area = 0.5 * (np.linalg.norm(np.cross(q[1] - q[0], q[2] - q[0])) + np.linalg.norm(np.cross(q[1] - q[3], q[2] - q[3])))
                chunk[t, 3] = self._area_light(first_prim + t, float(area), emission, emission_scale, two_sided)
        self.prims.append(chunk)
        self.owners.append(self._object)
        self._n_prims += n
        return first_prim

    def add_sphere(self, radius, material, render_from_object=None, reverse_orientation=False, z_min=None, z_max=None,
                   phi_max=360.0, emission=None, emission_scale=1.0, two_sided=False):
        rfo = IDENTITY if render_from_object is None else _as_f32(render_from_object, (4, 4))
        ofr = np.linalg.inv(rfo.astype(np.float64)).astype(np.float32)
        s = abi.ShmSphere()
        r = f32(radius)
        zmin = -r if z_min is None else f32(z_min)
        zmax = r if z_max is None else f32(z_max)
        s.radius = r
        s.z_min = float(np.clip(min(zmin, zmax), -r, r))
        s.z_max = float(np.clip(max(zmin, zmax), -r, r))
        s.theta_z_min = float(np.arccos(np.clip(f32(min(zmin, zmax) / r), -1, 1), dtype=np.float32))
        s.theta_z_max = float(np.arccos(np.clip(f32(max(zmin, zmax) / r), -1, 1), dtype=np.float32))
        s.phi_max = float(f32(np.pi / 180.0) * f32(np.clip(phi_max, 0, 360)))
        s.render_from_object[:] = [float(x) for x in rfo.ravel()]
        s.object_from_render[:] = [float(x) for x in ofr.ravel()]
        s.reverse_orientation = int(reverse_orientation)
        s.transform_swaps_handedness = int(np.linalg.det(rfo[:3, :3].astype(np.float64)) < 0)
        self.spheres.append(s)
        prim_index = self._n_prims
        li = -1
        if emission is not None:
            area = float(s.phi_max * s.radius * (s.z_max - s.z_min))
            li = self._area_light(prim_index, area, emission, emission_scale, two_sided)
        self.prims.append(np.array([[abi.SHM_SHAPE_SPHERE, len(self.spheres) - 1, material, li]], np.int64))
        self.owners.append(self._object)
        self._n_prims += 1
        return prim_index

    # ---- camera / film ----
    def set_film(self, width, height, pixel_bounds=None, filter_radius=(0.5, 0.5), imaging_ratio=1.0, max_component_value=np.inf):
        t = tables()
        self._sensor = [_as_f32(t["CIE_X"]), _as_f32(t["CIE_Y"]), _as_f32(t["CIE_Z"])]  # PixelSensor::new cie1931 (film.rs:823-837)
        f = abi.ShmFilm()
        pb = pixel_bounds if pixel_bounds is not None else (0, 0, width, height)
        f.pixel_bounds[:] = pb
        f.full_resolution[:] = (width, height)
        f.filter_radius[:] = filter_radius
        f.imaging_ratio = imaging_ratio
        f.max_component_value = max_component_value
        f.sensor_r_bar, f.sensor_g_bar, f.sensor_b_bar = (_fptr(a) for a in self._sensor)
        self.film = f

    def set_camera_look_at(self, lib, pos, look_at, up, fov, lens_radius=0.0, focal_distance=1e6, orthographic=False):
        """Transform::look_at (transform.rs:270-303, through shm_look_at) -> world_from_camera, then shm_camera_perspective.
        Returns render_from_world (4x4 f32) so that callers can move world-space geometry into render space."""
        wfc32 = np.zeros(16, np.float32)  # Transform::look_at in f32, exactly as the reference (and the C++ PBRT loader) computes it
        abi.check(lib, lib.shm_look_at(_fptr(_as_f32(pos)), _fptr(_as_f32(look_at)), _fptr(_as_f32(up)), _fptr(wfc32)), "shm_look_at")
        cam = abi.ShmCamera()
        rfw = np.zeros(16, np.float32)
        res = (C.c_int32 * 2)(*self.film.full_resolution)
        if orthographic:  # OrthographicCamera (camera.rs:658-840): `fov` is ignored, the screen window is [-aspect, aspect] x [-1, 1]
            abi.check(lib, lib.shm_camera_orthographic(_fptr(wfc32), res, float(lens_radius), float(focal_distance), C.byref(cam),
                                                       _fptr(rfw)), "shm_camera_orthographic")
        else:
            abi.check(lib, lib.shm_camera_perspective(_fptr(wfc32), float(fov), res, float(lens_radius), float(focal_distance),
                                                      C.byref(cam), _fptr(rfw)), "shm_camera_perspective")
        self.camera = cam
        return rfw.reshape(4, 4)

    # ---- finalize ----
    # ---- object instancing (loading/scene.rs:814-866; TransformedPrimitive, primitive.rs:136-176) ----
    def begin_object(self, name):
        """ObjectBegin: shapes added until end_object() belong to the object definition, not to the scene."""
        assert self._object == 0 and name not in self.objects
        self.objects[name] = len(self.objects) + 1
        self._object = self.objects[name]

    def end_object(self):
        self._object = 0

    def add_instance(self, name, render_from_instance=None):
        """ObjectInstance: one TransformedPrimitive of the named object's aggregate."""
        assert self._object == 0
        m = _as_f32(IDENTITY if render_from_instance is None else render_from_instance, (4, 4))
        self.instances.append((self.objects[name], m))
        self.prims.append(np.array([[abi.SHM_SHAPE_INSTANCE, len(self.instances) - 1, 0, -1]], np.int64))
        self.owners.append(0)
        self._n_prims += 1
        return self._n_prims - 1

    def prim_bounds(self):
        """Per-primitive Bounds3f in input order (triangle.rs:507-510; sphere.rs:275-280 + transform.rs:537-549)."""
        n = self._n_prims
        out = np.empty((n, 6), np.float32)
        src = np.concatenate(self.prims, axis=0)
        kinds, sidx = src[:, 0], src[:, 1]
        tri_mask = kinds == abi.SHM_SHAPE_TRIANGLE
        if tri_mask.any():
            allp = []
            for m in self.meshes:
                allp.append(m["p"][m["vi"].astype(np.int64)])  # (T,3,3)
            tri_p = np.concatenate(allp, axis=0)
            tp = tri_p[sidx[tri_mask]]
            out[tri_mask, :3] = tp.min(axis=1)
            out[tri_mask, 3:] = tp.max(axis=1)
        patch_mask = kinds == abi.SHM_SHAPE_BILINEAR_PATCH
        if patch_mask.any():  # bilinear_patch.rs:430-433: union of the four corners
            pp = np.concatenate([m["p"][m["vi"].astype(np.int64)] for m in self.patch_meshes], axis=0)[sidx[patch_mask]]  # (N,4,3)
            out[patch_mask, :3] = pp.min(axis=1)
            out[patch_mask, 3:] = pp.max(axis=1)
        for i in np.nonzero(kinds == abi.SHM_SHAPE_SPHERE)[0]:
            s = self.spheres[sidx[i]]
            m = np.asarray(list(s.render_from_object), np.float32).reshape(4, 4)
            r = f32(s.radius)
            lo, hi = np.array([-r, -r, s.z_min], np.float32), np.array([r, r, s.z_max], np.float32)
            pts = []
            for c in range(8):
                q = np.array([hi[0] if c & 1 else lo[0], hi[1] if c & 2 else lo[1], hi[2] if c & 4 else lo[2]], np.float32)
                xp = f32(f32(f32(m[0, 0] * q[0]) + f32(m[0, 1] * q[1])) + f32(m[0, 2] * q[2])) + m[0, 3]
                yp = f32(f32(f32(m[1, 0] * q[0]) + f32(m[1, 1] * q[1])) + f32(m[1, 2] * q[2])) + m[1, 3]
                zp = f32(f32(f32(m[2, 0] * q[0]) + f32(m[2, 1] * q[1])) + f32(m[2, 2] * q[2])) + m[2, 3]
                pts.append([f32(xp), f32(yp), f32(zp)])
            pts = np.asarray(pts, np.float32)
            out[i, :3], out[i, 3:] = pts.min(axis=0), pts.max(axis=0)
        return out

    def build(self, lib, split_method=0):
        """BvhAggregate::new + marshal. Returns (ShmSceneDesc, info dict). Keeps all arrays alive on self."""
        assert self.camera is not None and self.film is not None
        n = self._n_prims
        bounds = np.ascontiguousarray(self.prim_bounds())
        src = np.concatenate(self.prims, axis=0)
        owner = np.concatenate([np.full(len(c), o, np.int64) for c, o in zip(self.prims, self.owners)])
        node_dt = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("n_prims", "<u2"), ("axis", "u1"), ("pad", "u1")])

        def build_tree(idx):
            """BvhAggregate::new over the primitives idx (input indices): (node records, idx in leaf order)."""
            b = np.ascontiguousarray(bounds[idx])
            nd = (abi.ShmBvhNode * (2 * len(idx)))()
            order = np.zeros(len(idx), np.uint32)
            cnt = C.c_uint32(0)
            abi.check(lib, lib.shm_bvh_build(_fptr(b), len(idx), split_method, nd, C.byref(cnt), order.ctypes.data_as(abi.c_u32_p)), "shm_bvh_build")
            return np.frombuffer(nd, dtype=node_dt)[:cnt.value].copy(), idx[order]

        # the aggregate of every instanced object first (their root bounds give the instances' bounds), then the scene's own tree
        obj_trees = {}
        for k in sorted(set(self.objects.values())):
            idx = np.nonzero(owner == k)[0]
            assert len(idx) > 0, "empty object definition"
            obj_trees[k] = build_tree(idx)
        for ii, (k, m) in enumerate(self.instances):  # Transform::apply(Bounds3f), transform.rs:557-571: the eight corners
            root = obj_trees[k][0][0]
            lo, hi = root["bmin"], root["bmax"]
            pts = []
            for c in range(8):
                q = np.array([hi[0] if c & 1 else lo[0], hi[1] if c & 2 else lo[1], hi[2] if c & 4 else lo[2]], np.float32)
                pts.append([f32(f32(f32(f32(m[r, 0] * q[0]) + f32(m[r, 1] * q[1])) + f32(m[r, 2] * q[2])) + m[r, 3]) for r in range(3)])
            pts = np.asarray(pts, np.float32)
            slot = np.nonzero((src[:, 0] == abi.SHM_SHAPE_INSTANCE) & (src[:, 1] == ii))[0][0]
            bounds[slot, :3], bounds[slot, 3:] = pts.min(axis=0), pts.max(axis=0)
        top_nodes, top_order = build_tree(np.nonzero(owner == 0)[0])
        trees = [(top_nodes, top_order)] + [obj_trees[k] for k in sorted(obj_trees)]
        node_base, prim_base, nb, pb = {}, {}, 0, 0
        for t, k in zip(trees, [0] + sorted(obj_trees)):
            node_base[k], prim_base[k] = nb, pb
            leaf = t[0]["n_prims"] > 0
            t[0]["offset"][leaf] += pb
            t[0]["offset"][~leaf] += nb
            nb += len(t[0])
            pb += len(t[1])
        all_nodes = np.concatenate([t[0] for t in trees])
        order = np.concatenate([t[1] for t in trees]).astype(np.uint32)  # input index of every leaf-order slot
        nodes = (abi.ShmBvhNode * len(all_nodes)).from_buffer_copy(all_nodes.tobytes())
        n_nodes = C.c_uint32(len(all_nodes))
        slot_of_input = np.empty(n, np.uint32)
        slot_of_input[order] = np.arange(n, dtype=np.uint32)
        prim_arr = (abi.ShmPrimitive * n)()
        pa = np.frombuffer(prim_arr, dtype=np.dtype([("k", "<u4"), ("i", "<u4"), ("m", "<u4"), ("l", "<i4")]))
        pa["k"], pa["i"], pa["m"], pa["l"] = src[order, 0], src[order, 1], src[order, 2], src[order, 3]
        instances = (abi.ShmInstance * max(1, len(self.instances)))()
        for ii, (k, m) in enumerate(self.instances):
            instances[ii].render_from_primitive[:] = [float(x) for x in m.ravel()]
            instances[ii].primitive_from_render[:] = [float(x) for x in _as_f32(np.linalg.inv(m.astype(np.float64))).ravel()]
            instances[ii].root_node = node_base[k]
        lights = (abi.ShmLight * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            lights[i] = l
            if l.kind == abi.SHM_LIGHT_DIFFUSE_AREA:
                lights[i].primitive = int(slot_of_input[l.primitive])
        meshes = (abi.ShmTriangleMesh * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            mm = meshes[i]
            mm.n_triangles, mm.n_vertices = m["vi"].shape[0], m["p"].shape[0]
            mm.vertex_indices = m["vi"].ctypes.data_as(abi.c_u32_p)
            mm.p = _fptr(m["p"])
            mm.n = _fptr(m["n"]) if m["n"] is not None else None
            mm.s = _fptr(m["s"]) if m["s"] is not None else None
            mm.uv = _fptr(m["uv"]) if m["uv"] is not None else None
            mm.reverse_orientation, mm.transform_swaps_handedness = int(m["reverse"]), int(m["swaps"])
        spheres = (abi.ShmSphere * max(1, len(self.spheres)))(*self.spheres)
        patch_meshes = (abi.ShmBilinearPatchMesh * max(1, len(self.patch_meshes)))()
        for i, m in enumerate(self.patch_meshes):
            pm = patch_meshes[i]
            pm.n_patches, pm.n_vertices = m["vi"].shape[0], m["p"].shape[0]
            pm.vertex_indices = m["vi"].ctypes.data_as(abi.c_u32_p)
            pm.p = _fptr(m["p"])
            pm.n = _fptr(m["n"]) if m["n"] is not None else None
            pm.uv = _fptr(m["uv"]) if m["uv"] is not None else None
            pm.reverse_orientation, pm.transform_swaps_handedness = int(m["reverse"]), int(m["swaps"])
        materials = (abi.ShmMaterial * len(self.materials))(*self.materials)
        spec = np.concatenate(self.spec).astype(np.float32) if self.spec else np.zeros(1, np.float32)
        d = abi.ShmSceneDesc()
        d.abi_version = abi.SHM_ABI_VERSION
        d.n_nodes, d.nodes = n_nodes.value, nodes
        d.n_primitives, d.primitives = n, prim_arr
        d.n_meshes, d.meshes = len(self.meshes), meshes
        d.n_spheres, d.spheres = len(self.spheres), spheres
        d.n_materials, d.materials = len(self.materials), materials
        d.n_lights, d.lights = len(self.lights), lights
        d.n_spectrum_floats, d.spectrum_data = spec.size, _fptr(spec)
        d.camera, d.film = self.camera, self.film
        d.n_patch_meshes, d.patch_meshes = len(self.patch_meshes), patch_meshes
        d.n_instances, d.instances = len(self.instances), instances
        self._keep = [nodes, prim_arr, lights, meshes, spheres, materials, spec, bounds, order, patch_meshes, instances]
        if self.textures or self.image_lights:
            textures = (abi.ShmImageTexture * max(1, len(self.textures)))(*self.textures)
            levels = (abi.ShmImageLevel * len(self.tex_levels))()
            for i, (w, h, off) in enumerate(self.tex_levels):
                levels[i].width, levels[i].height, levels[i].texel_offset = w, h, off
            texels = np.concatenate(self.texels).astype(np.float32)
            lut = _as_f32(tables()["MIP_FILTER_LUT"])
            d.n_image_textures, d.image_textures = len(self.textures), textures
            d.n_image_levels, d.image_levels = len(self.tex_levels), levels
            d.n_texel_floats, d.texel_data = texels.size, _fptr(texels)
            d.ewa_filter_lut = _fptr(lut)
            self._keep += [textures, levels, texels, lut]
            if self.color_space is not None:
                cs = self.color_space
                d.color_space.rgb2spec_res = cs["res"]
                d.color_space.rgb2spec_scale, d.color_space.rgb2spec_data = _fptr(cs["scale"]), _fptr(cs["data"])
                d.color_space.illuminant = _fptr(cs["illuminant"])
            if self.image_lights:
                ils = (abi.ShmImageInfiniteLight * len(self.image_lights))(*self.image_lights)
                d.n_image_lights, d.image_lights = len(self.image_lights), ils
                self._keep.append(ils)
        if self.float_textures:
            fts = (abi.ShmFloatTexture * len(self.float_textures))(*self.float_textures)
            d.n_float_textures, d.float_textures = len(self.float_textures), fts
            self._keep.append(fts)
        if self.spectrum_textures:
            sts = (abi.ShmSpectrumTexture * len(self.spectrum_textures))(*self.spectrum_textures)
            d.n_spectrum_textures, d.spectrum_textures = len(self.spectrum_textures), sts
            self._keep.append(sts)
        info = dict(n_nodes=n_nodes.value, n_primitives=n, order=order, slot_of_input=slot_of_input, bounds=bounds)
        return d, info


def tiles_for(lib, pixel_bounds, tile=8):
    """Tile::tile(pixel_bounds, 8, 8) through the host mirror."""
    pb = (C.c_int32 * 4)(*pixel_bounds)
    w, h = pixel_bounds[2] - pixel_bounds[0], pixel_bounds[3] - pixel_bounds[1]
    cap = ((w + tile - 1) // tile) * ((h + tile - 1) // tile)
    tiles = (abi.ShmTile * cap)()
    n = C.c_uint32(0)
    abi.check(lib, lib.shm_tile_bounds(pb, tile, tile, tiles, C.byref(n)), "shm_tile_bounds")
    return tiles, n.value


def wave_schedule(spp):
    """integrator.rs:231-233, 306-308: (start, end) of each spp-wave: sizes 1,1,2,4,...,64,64,..."""
    waves, ws, we, nxt = [], 0, 1, 1
    while ws < spp:
        waves.append((ws, we))
        ws = we
        we = min(spp, we + nxt)
        nxt = min(2 * nxt, 64)
    return waves
