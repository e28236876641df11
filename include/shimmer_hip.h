/* shimmer_hip.h — C ABI of libshimmer_hip.so: the MI355X (gfx950) wavefront path tracer that drops in
 * behind Shimmer's `Integrator::render()`.
 *
 * The reference has no FFI/plugin interface; its only seam is
 *     pub trait Integrator { fn render(&mut self, options: &Options); }      (src/integrator.rs:52-54)
 * constructed by create_integrator(name, params, camera, sampler, aggregate, lights, color_space)
 * (src/integrator.rs:16-42) and invoked once from render_cpu (src/render.rs:51-54).  A GPU backend is a
 * new `impl Integrator` whose render() marshals the already-built scene objects (flattened BVH of
 * src/aggregate.rs:471-481, triangle meshes of src/shape/mesh.rs:9-20, materials, lights, camera, film
 * sensor) into the flat POD arrays below and calls shm_scene_create + shm_render_wave per spp-wave
 * (src/integrator.rs:241-320), then reads the film sums back.  INTEGRATION.md shows that Rust shim.
 *
 * Conventions: every function returns 0 (SHM_OK) or a negative ShmError; nothing throws, aborts or
 * unwinds across this boundary (the reference panics; a Rust caller maps codes to Result).  All pointers
 * are borrowed for the duration of the call only; the library copies what it keeps.  Everything is
 * little-endian f32/u32/i32 unless stated.  One in-flight render per ShmScene; distinct scenes may be
 * used from distinct host threads.  No callbacks into the host.
 */
#ifndef SHIMMER_HIP_H
#define SHIMMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHM_ABI_VERSION 8

/* The library is built with -fvisibility=hidden; only these entry points are exported. */
#if defined(__GNUC__)
#define SHM_API __attribute__((visibility("default")))
#else
#define SHM_API
#endif

typedef enum ShmError {
    SHM_OK = 0,
    SHM_ERR_INVALID_ARGUMENT = -1,
    SHM_ERR_UNSUPPORTED = -2,   /* valid input the backend does not take (nested instances, trees over the node limits, BVH deeper than 64, ...) */
    SHM_ERR_DEVICE = -3,        /* HIP runtime error; see shm_last_error() */
    SHM_ERR_NO_DEVICE = -4,     /* no gfx950 device visible: there is NO CPU fallback */
    SHM_ERR_OUT_OF_MEMORY = -5,
    SHM_ERR_INTERNAL = -6
} ShmError;

/* ---- geometry -------------------------------------------------------------------------------- */

/* LinearBvhNode (src/aggregate.rs:471-481; 64 B in the reference) narrowed to 32 B.
 * DFS order: first child of interior node i is i+1; `offset` is second_child_offset (interior) or
 * primitive_offset into the leaf-ordered primitive list (leaf); n_prims > 0 marks a leaf. */
typedef struct ShmBvhNode {
    float bmin[3];
    float bmax[3];
    uint32_t offset;
    uint16_t n_prims;
    uint8_t axis;
    uint8_t pad;
} ShmBvhNode;

/* TriangleMesh (src/shape/mesh.rs:9-20), vertices already in render space (mesh.rs:43-46). */
typedef struct ShmTriangleMesh {
    uint32_t n_triangles;
    uint32_t n_vertices;
    const uint32_t* vertex_indices; /* 3*n_triangles (the reference's Vec<usize>, narrowed) */
    const float* p;                 /* 3*n_vertices */
    const float* n;                 /* 3*n_vertices or NULL */
    const float* s;                 /* 3*n_vertices or NULL */
    const float* uv;                /* 2*n_vertices or NULL */
    uint8_t reverse_orientation;
    uint8_t transform_swaps_handedness;
    uint8_t pad[6];
} ShmTriangleMesh;

/* BilinearPatchMesh (src/shape/mesh.rs:289-376) for BilinearPatch (src/shape/bilinear_patch.rs), vertices in render space.
 * (Image-valued emission on a patch light is not carried: `uv` only drives the (s, t) parameterisation of the interaction.) */
typedef struct ShmBilinearPatchMesh {
    uint32_t n_patches;
    uint32_t n_vertices;
    const uint32_t* vertex_indices; /* 4*n_patches: p00, p10, p01, p11 (bilinear_patch.rs:87-98) */
    const float* p;                 /* 3*n_vertices */
    const float* n;                 /* 3*n_vertices or NULL */
    const float* uv;                /* 2*n_vertices or NULL */
    uint8_t reverse_orientation;
    uint8_t transform_swaps_handedness;
    uint8_t pad[6];
} ShmBilinearPatchMesh;

/* Sphere (src/shape/sphere.rs:27-38); matrices row-major m[r][c] as SquareMatrix<4>. */
typedef struct ShmSphere {
    float radius, z_min, z_max, theta_z_min, theta_z_max, phi_max;
    float render_from_object[16];
    float object_from_render[16];
    uint8_t reverse_orientation;
    uint8_t transform_swaps_handedness;
    uint8_t pad[6];
} ShmSphere;

enum { SHM_SHAPE_TRIANGLE = 0, SHM_SHAPE_SPHERE = 1, SHM_SHAPE_BILINEAR_PATCH = 2,
       SHM_SHAPE_INSTANCE = 3 /* ABI v6: a TransformedPrimitive (primitive.rs:136-176): shape_index = index into ShmSceneDesc::instances;
                                 its material / area_light fields are ignored (the instanced primitives carry their own) */ };
/* TransformedPrimitive {primitive, render_from_primitive}: the instanced primitive is a BvhAggregate of the object's shapes
 * (loading/scene.rs:817-829) stored as ANOTHER tree in ShmSceneDesc::nodes starting at root_node (DFS order and absolute
 * offsets as for the top-level tree at node 0; its leaves index ShmSceneDesc::primitives like every other leaf). An object with a
 * single shape is a one-leaf tree. Constraints: the instance primitive is alone in its top-level leaf (the reference's builder puts
 * one primitive per leaf unless centroids coincide); instanced primitives are not instances and carry no area light.
 * Reference behaviour kept as written: intersect() maps the ray with apply_ray_inverse, intersect_predicate() with the FORWARD
 * apply_ray (primitive.rs:158-175), and Transform::apply(SurfaceInteraction) maps every vector with the inverse (quirk 6). */
typedef struct ShmInstance {
    float render_from_primitive[16];  /* matrix m, row-major */
    float primitive_from_render[16];  /* m_inv */
    uint32_t root_node;
    uint32_t pad[3];
} ShmInstance;

/* GeometricPrimitive / SimplePrimitive (src/primitive.rs:66-130): shape + material (+ area light).
 * Listed in the order of BvhAggregate::primitives AFTER the build's reordering (aggregate.rs:270),
 * i.e. leaf node `offset` indexes this array directly. */
typedef struct ShmPrimitive {
    uint32_t shape_kind;   /* SHM_SHAPE_* */
    uint32_t shape_index;  /* triangle: global triangle index (mesh base + tri_index); sphere: index;
                              bilinear patch: global patch index (patch-mesh base + blp_index) */
    uint32_t material;     /* index into materials. 0xffffffff (the reference's `material: None`, a medium interface that li() skips with
                              skip_intersection, interaction.rs:410-427) is rejected with SHM_ERR_UNSUPPORTED: media are todo!() there */
    int32_t area_light;    /* index into lights, or -1 */
} ShmPrimitive;

/* ---- spectra, materials, lights -------------------------------------------------------------- */

enum {
    SHM_SPECTRUM_CONSTANT = 0,         /* ConstantSpectrum          (spectra/spectrum.rs:143-166) */
    SHM_SPECTRUM_DENSE = 1,            /* DenselySampledSpectrum, lambda_min..=lambda_max, 1 nm (:168-291) */
    SHM_SPECTRUM_PIECEWISE_LINEAR = 2, /* PiecewiseLinearSpectrum   (:293-428): n lambdas then n values */
    /* RGB-derived spectra: the host looks the sigmoid coefficients up (RgbColorSpace::to_rgb_coeffs, colorspace.rs:95) and
     * hands them over; the device evaluates RgbSigmoidPolynomial::get = s(c0 l^2 + c1 l + c2) (color.rs:333-383). */
    SHM_SPECTRUM_RGB_ALBEDO = 3,       /* RgbAlbedoSpectrum     (:498-528) */
    SHM_SPECTRUM_RGB_UNBOUNDED = 4,    /* RgbUnboundedSpectrum  (:531-565): c = scale (2 max(r,g,b)) */
    SHM_SPECTRUM_RGB_ILLUMINANT = 5,   /* RgbIlluminantSpectrum (:568-607): c = scale; offset / n / lambda_min: the colour space's
                                          illuminant as a densely sampled table */
    /* ABI v6. A SpectrumImageTexture (texture.rs:689-808) bound to a material's SpectrumTexture slot (ShmMaterial a, b, c):
     * `offset` = index into ShmSceneDesc::image_textures. Not valid for eta (a Spectrum, not a texture, in the reference) nor
     * for lights. The device filters the MIP pyramid, looks the sigmoid coefficients up in ShmSceneDesc::color_space and
     * evaluates the resulting Rgb{Albedo,Unbounded,Illuminant}Spectrum at the path's wavelengths. */
    SHM_SPECTRUM_IMAGE_TEXTURE = 6,
    /* ABI v6. A composite SpectrumTexture (scale / mix / directionmix, texture.rs:536-687, 810-828) bound to a material's
     * SpectrumTexture slot: `offset` = index into ShmSceneDesc::spectrum_textures (the root of its tree). */
    SHM_SPECTRUM_TEXTURE_NODE = 7
};
typedef struct ShmSpectrum {
    uint32_t kind;
    float c;              /* CONSTANT */
    uint32_t offset;      /* DENSE / PIECEWISE: first float in ShmSceneDesc::spectrum_data */
    uint32_t n;           /* DENSE: number of samples; PIECEWISE: number of knots */
    int32_t lambda_min;   /* DENSE */
    float rgb_c[3];       /* RGB_*: c0, c1, c2 of the sigmoid polynomial */
} ShmSpectrum;

enum {
    SHM_MATERIAL_DIFFUSE = 0,         /* material.rs:247-332 */
    SHM_MATERIAL_CONDUCTOR = 1,       /* material.rs:334-516 */
    SHM_MATERIAL_DIELECTRIC = 2,      /* material.rs:518-650 */
    SHM_MATERIAL_THIN_DIELECTRIC = 3, /* material.rs:652-768 */
    SHM_MATERIAL_COATED_DIFFUSE = 4,  /* material.rs:772-1005: LayeredBxDF<Dielectric, Diffuse, two-sided> (bxdf.rs:269-290, 883-1620) */
    SHM_MATERIAL_COATED_CONDUCTOR = 5,/* material.rs:1007-1286: LayeredBxDF<Dielectric, Conductor, two-sided> (bxdf.rs:460-480) */
    SHM_MATERIAL_MIX = 6              /* material.rs:1288-1330: MixMaterial, resolved per hit in get_bsdf (interaction.rs:205-220).
                                         The reference draws the choice from the tile's entropy-seeded SmallRng (integrator.rs:255);
                                         here it is a hash of (wo, p), the way PBRT-v4 defines it: reproducible, parity unpinned. */
};
/* FloatTexture (texture.rs:88-305, 309-403), ABI v6: a node table; children are indices into ShmSceneDesc::float_textures and must
 * precede their parent (no cycles); a tree (counting a shared child once per use) may hold at most 32 nodes. Materials refer to a node through ShmMaterial::float_tex. */
enum {
    SHM_FLOATTEX_CONSTANT = 0,      /* FloatConstantTexture: value */
    SHM_FLOATTEX_SCALED = 1,        /* FloatScaledTexture:  tex = a, scale = b */
    SHM_FLOATTEX_MIX = 2,           /* FloatMixTexture:     tex1 = a, tex2 = b, amount = c */
    SHM_FLOATTEX_DIRECTION_MIX = 3, /* FloatDirectionMixTexture: tex1 = a, tex2 = b, dir */
    SHM_FLOATTEX_IMAGE = 4          /* FloatImageTexture: image = index into ShmSceneDesc::image_textures (its spectrum_type and
                                       has_color_space are ignored) */
};
typedef struct ShmFloatTexture {
    uint32_t kind;
    float value;
    uint32_t a, b, c;
    float dir[3];
    uint32_t image;
    uint32_t pad[3];
} ShmFloatTexture;
/* SpectrumTexture node table (texture.rs:411-503): a LEAF is a spectrum or an image texture (SpectrumConstantTexture /
 * SpectrumImageTexture); composites name their children by index (children precede parents; a tree holds at most 8 nodes) and
 * their FloatTexture parameter by index into ShmSceneDesc::float_textures. */
enum {
    SHM_SPECTEX_LEAF = 0,          /* leaf: any ShmSpectrum kind except TEXTURE_NODE */
    SHM_SPECTEX_SCALED = 1,        /* SpectrumScaledTexture: tex = a, scale = float texture f */
    SHM_SPECTEX_MIX = 2,           /* SpectrumMixTexture: tex1 = a, tex2 = b, amount = float texture f */
    SHM_SPECTEX_DIRECTION_MIX = 3  /* SpectrumDirectionMixTexture: tex1 = a, tex2 = b, dir */
};
typedef struct ShmSpectrumTexture {
    uint32_t kind;
    uint32_t a, b;
    uint32_t f;
    float dir[3];
    uint32_t pad;
    ShmSpectrum leaf;
} ShmSpectrumTexture;
/* indices into ShmMaterial::float_tex */
enum {
    SHM_FLOATSLOT_DISPLACEMENT = 0, SHM_FLOATSLOT_U_ROUGHNESS = 1, SHM_FLOATSLOT_V_ROUGHNESS = 2, SHM_FLOATSLOT_U2_ROUGHNESS = 3,
    SHM_FLOATSLOT_V2_ROUGHNESS = 4, SHM_FLOATSLOT_THICKNESS = 5, SHM_FLOATSLOT_G = 6, SHM_FLOATSLOT_MIX_AMOUNT = 7
};
/* Every float parameter is a constant unless float_tex[slot] != 0; spectrum slots a, b, c may bind image textures. */
typedef struct ShmMaterial {
    uint32_t kind;
    uint32_t has_displacement; /* Diffuse: always 1 with a constant-0 texture (material.rs:280, quirk 8) */
    float displacement;        /* the constant displacement value */
    uint32_t remap_roughness;
    float u_roughness, v_roughness; /* Conductor / Dielectric; Coated*: the dielectric interface */
    float u2_roughness, v2_roughness; /* CoatedConductor: the conductor (material.rs:1237-1244 derives it from the interface
                                         roughness when remap_roughness is set: reference behaviour, done in get_bsdf) */
    float thickness, g;               /* Coated*: layer thickness, HG asymmetry of the medium between the interfaces */
    int32_t max_depth, n_samples;     /* Coated*: random-walk depth and walks per evaluation (defaults 10, 1) */
    uint32_t conductor_from_reflectance; /* CoatedConductor: `a` is a reflectance (material.rs:1224-1231), not eta */
    uint32_t mix_material[2];         /* Mix: indices into the material table (may themselves be Mix; no cycles) */
    float mix_amount;                 /* Mix: constant `amount` texture (default 0.5): <= 0 -> [0], >= 1 -> [1], else [amount < u ? 0 : 1] */
    ShmSpectrum a;  /* Diffuse / CoatedDiffuse: reflectance; Conductor / CoatedConductor: eta (or reflectance); Dielectric/Thin: eta */
    ShmSpectrum b;  /* Conductor / CoatedConductor: k */
    ShmSpectrum c;  /* Coated*: albedo of the medium */
    ShmSpectrum d;  /* Coated*: eta of the dielectric interface */
    uint32_t float_tex[8];  /* ABI v6: SHM_FLOATSLOT_*: 0 = the constant field above, else 1 + index into ShmSceneDesc::float_textures */
    uint32_t normal_map;    /* 0 = none, else 1 + index into ShmSceneDesc::image_textures of the normal map (3 channels; only its finest
                               level, repeat wrap and bilinear lookup are used: material::normal_map, material.rs:1453-1475). Ignored
                               when has_displacement is set (interaction.rs:223-236), so never reached on a DiffuseMaterial (quirk 8) */
    uint32_t pad[3];
} ShmMaterial;

enum {
    SHM_LIGHT_POINT = 0,            /* light.rs:392-497 */
    SHM_LIGHT_DIFFUSE_AREA = 1,     /* light.rs:499-690; one per emissive shape (loading/scene.rs:609-624) */
    SHM_LIGHT_UNIFORM_INFINITE = 2, /* light.rs:692-816 */
    SHM_LIGHT_IMAGE_INFINITE = 3    /* ImageInfinitelight, light.rs:805-981 (ABI v6): `primitive` = index into ShmSceneDesc::image_lights;
                                       `scale` as for the others; `spectrum` unused (the radiance is the image's RGB as an
                                       RgbIlluminantSpectrum of ShmSceneDesc::color_space, which must carry table and illuminant) */
};
typedef struct ShmLight {
    uint32_t kind;
    uint32_t primitive;    /* DIFFUSE_AREA: index into primitives (leaf order) of its shape */
    float scale;           /* final scale (after /spectrum_to_photometric, power...) */
    uint32_t two_sided;
    float position[3];     /* POINT: render_from_light(0,0,0) */
    float area;            /* DIFFUSE_AREA: shape.area() (light.rs:543) */
    ShmSpectrum spectrum;  /* DenselySampledSpectrum of I / Lemit (must be DENSE) */
} ShmLight;

/* ---- camera, film ---------------------------------------------------------------------------- */

enum {
    SHM_CAMERA_PERSPECTIVE = 0,  /* camera.rs:866-1079 */
    SHM_CAMERA_ORTHOGRAPHIC = 1  /* camera.rs:658-840: rays start at camera_from_raster(p_film) and run along +z; the reference
                                    has no depth of field for it yet ("TODO Adjust for depth-of-field here", :762) */
};
/* ---- image textures (ABI v6) ------------------------------------------------------------------ */

/* One level of a MIPMap pyramid (Image::generate_pyramid, image.rs:699-800): `width * height * n_channels` floats at
 * ShmSceneDesc::texel_data + texel_offset, row-major from the TOP row, channels interleaved, each value what
 * Image::get_channel returns for that texel (image.rs:452-476: u8 through the colour encoding, f16 widened, f32 as is). */
typedef struct ShmImageLevel {
    int32_t width, height;
    uint32_t texel_offset;
    uint32_t pad;
} ShmImageLevel;
enum { SHM_TEXMAP_UV = 0, SHM_TEXMAP_SPHERICAL = 1, SHM_TEXMAP_CYLINDRICAL = 2, SHM_TEXMAP_PLANAR = 3 }; /* texture.rs:838-843 */
enum { SHM_TEXFILTER_POINT = 0, SHM_TEXFILTER_BILINEAR = 1, SHM_TEXFILTER_TRILINEAR = 2, SHM_TEXFILTER_EWA = 3 }; /* mipmap.rs:334-341 */
enum { SHM_WRAP_BLACK = 0, SHM_WRAP_CLAMP = 1, SHM_WRAP_REPEAT = 2, SHM_WRAP_OCTAHEDRAL_SPHERE = 3 };  /* image.rs:73-78 */
enum { SHM_SPECTRUM_TYPE_ALBEDO = 0, SHM_SPECTRUM_TYPE_UNBOUNDED = 1, SHM_SPECTRUM_TYPE_ILLUMINANT = 2 };
/* SpectrumImageTexture = ImageTextureBase {mapping, scale, invert, mipmap} + spectrum_type (texture.rs:19-26, 689-693);
 * MIPMap {pyramid, color_space, wrap_mode, options} (mipmap.rs:8-14). */
typedef struct ShmImageTexture {
    uint32_t mapping;            /* SHM_TEXMAP_* */
    float su, sv, du, dv;        /* UVMapping (texture.rs:896-905); PlanarMapping: du, dv are its ds, dt */
    float vs[3], vt[3];          /* PlanarMapping (texture.rs:1012-1019) */
    float texture_from_render[16]; /* Spherical / Cylindrical / Planar: the matrix m of texture_from_render, row-major */
    uint32_t filter;             /* SHM_TEXFILTER_* ("filter", default bilinear, texture.rs:738) */
    float max_anisotropy;        /* "maxanisotropy", default 8 */
    uint32_t wrap;               /* SHM_WRAP_* ("wrap", default repeat) */
    float scale;                 /* "scale", default 1 */
    uint8_t invert;
    uint8_t spectrum_type;       /* SHM_SPECTRUM_TYPE_* */
    uint8_t n_channels;          /* 1 or 3 (an RGBA image hands over its RGB: texel_rgb / bilerp read channels 0..2, mipmap.rs:204-231) */
    uint8_t has_color_space;     /* MIPMap::get_color_space().is_some(); 0: one-channel texture -> constant spectrum (texture.rs:801-805) */
    uint32_t first_level;        /* index into ShmSceneDesc::image_levels, finest level first */
    uint32_t n_levels;
} ShmImageTexture;
/* What RgbColorSpace::to_rgb_coeffs reads (colorspace.rs:95-98 -> rgb_to_spectra.rs:16-25): the rgb2spec coefficient table of
 * the colour space's gamut (the `.spec` file the reference loads: res, scale[res], data[3][res][res][res][3]) and its
 * illuminant (for SHM_SPECTRUM_TYPE_ILLUMINANT). One colour space per scene. */
typedef struct ShmColorSpace {
    uint32_t rgb2spec_res;       /* 0: no table (then no texture may have has_color_space) */
    uint32_t pad;
    const float* rgb2spec_scale; /* res floats */
    const float* rgb2spec_data;  /* 3 * res^3 * 3 floats */
    const float* illuminant;     /* DenselySampledSpectrum 360..=830, 471 floats (may be NULL without ILLUMINANT textures) */
} ShmColorSpace;

/* ImageInfinitelight (light.rs:805-816, 915-981): the environment map and the light's transform. The library derives both sampling
 * distributions from the image exactly as ImageInfinitelight::new does (Image::get_default_sampling_distribution = the channel
 * average per pixel, image.rs:1379-1405; PiecewiseConstant2D::new, sampling.rs:124-153; the compensated one with the average
 * subtracted, light.rs:948-955). */
typedef struct ShmImageInfiniteLight {
    float render_from_light[16];  /* matrix m, row-major */
    float light_from_render[16];  /* its inverse m_inv */
    uint32_t image_level;         /* index into ShmSceneDesc::image_levels: a square image, 3 channels, equal-area octahedral layout */
    uint32_t pad;
} ShmImageInfiniteLight;

/* PerspectiveCamera / OrthographicCamera after construction (ProjectiveCameraBase, camera.rs:594-642). Matrices row-major. */
typedef struct ShmCamera {
    float camera_from_raster[16];
    float render_from_camera[16];
    float dx_camera[3];
    float dy_camera[3];
    float lens_radius;
    float focal_distance;
    float shutter_open, shutter_close;
    uint32_t kind;                /* SHM_CAMERA_* (ABI v4) */
    uint32_t pad;
    /* ABI v6, read only by scenes with image textures (the ray-differential fallback Camera::approximate_dp_dxy, camera.rs:307-354):
     * the inverse matrix of render_from_camera and the four vectors CameraBase::find_minimum_differentials leaves (camera.rs:356-440).
     * shm_camera_perspective / shm_camera_orthographic fill them. */
    float camera_from_render[16];
    float min_pos_differential_x[3], min_pos_differential_y[3];
    float min_dir_differential_x[3], min_dir_differential_y[3];
} ShmCamera;

/* RgbFilm + PixelSensor (film.rs:470-574, 754-914) + BoxFilter radius (filter.rs:61-105). */
typedef struct ShmFilm {
    int32_t pixel_bounds[4];      /* min.x, min.y, max.x, max.y (max exclusive) */
    int32_t full_resolution[2];
    float filter_radius[2];
    float imaging_ratio;
    float max_component_value;
    const float* sensor_r_bar;    /* DenselySampledSpectrum 360..=830, 471 floats */
    const float* sensor_g_bar;
    const float* sensor_b_bar;
} ShmFilm;

typedef struct ShmSceneDesc {
    uint32_t abi_version;         /* SHM_ABI_VERSION */
    uint32_t n_nodes;
    const ShmBvhNode* nodes;
    uint32_t n_primitives;
    const ShmPrimitive* primitives;
    uint32_t n_meshes;
    const ShmTriangleMesh* meshes;
    uint32_t n_spheres;
    const ShmSphere* spheres;
    uint32_t n_materials;
    const ShmMaterial* materials;
    uint32_t n_lights;
    const ShmLight* lights;       /* the integrator's `lights` vector, in order (light_sampler.rs:83-103) */
    uint32_t n_spectrum_floats;
    const float* spectrum_data;
    ShmCamera camera;
    ShmFilm film;
    uint32_t n_patch_meshes;      /* ABI v3 */
    uint32_t pad;
    const ShmBilinearPatchMesh* patch_meshes;
    /* ABI v6: image textures */
    uint32_t n_image_textures;
    uint32_t n_image_levels;
    const ShmImageTexture* image_textures;
    const ShmImageLevel* image_levels;
    uint64_t n_texel_floats;
    const float* texel_data;
    ShmColorSpace color_space;
    const float* ewa_filter_lut;  /* MIP_FILTER_LUT (mipmap.rs:390-521), 128 floats; required when a texture uses SHM_TEXFILTER_EWA */
    uint32_t n_image_lights;
    uint32_t n_float_textures;
    const ShmImageInfiniteLight* image_lights;
    const ShmFloatTexture* float_textures;
    uint32_t n_spectrum_textures;
    uint32_t n_instances;
    const ShmSpectrumTexture* spectrum_textures;
    const ShmInstance* instances;
} ShmSceneDesc;

/* ---- render parameters ----------------------------------------------------------------------- */

/* options.rs:15-61 (the five flags the path reads) + PathIntegrator params (integrator.rs:188-192)
 * + sampler params (sampler.rs:95-99). */
typedef struct ShmRenderParams {
    uint64_t seed;
    int32_t samples_per_pixel;    /* total spp (ray-differential scale, integrator.rs:356-359) */
    int32_t max_depth;            /* "maxdepth", default 5 */
    uint8_t regularize;
    uint8_t disable_pixel_jitter;
    uint8_t disable_wavelength_jitter;
    uint8_t force_diffuse;        /* options.force_diffuse: every BSDF becomes DiffuseBxDF(rho_hd estimate), interaction.rs:256-275 */
    uint8_t integrator;           /* SHM_INTEGRATOR_* (ABI v5); 0 = "path" */
    uint8_t sample_lights;        /* SimplePath "samplelights" (default true in the reference, integrator.rs:135-137) */
    uint8_t sample_bsdf;          /* SimplePath "samplebsdf"   (default true) */
    uint8_t disable_texture_filtering; /* options.disable_texture_filtering: compute_differentials leaves zeros (interaction.rs:287-295) */
    /* ABI v8 — SHM_REFERENCE_QUIRKS. 0 (the default) = reference-exact: every deviation of the reference from PBRT-v4 that SURVEY 7
     * lists is reproduced, and every parity test runs this way. 1 = the PBRT-v4 behaviour at the sites where the reference's produces invalid or
     * biased values: LayeredBxDF::pdf tests the reflected sample (`rs.f != 0 && rs.pdf > 0`, bxdf.rs:1491-1506 omits it: 0/0 -> NaN film
     * pixels in coated scenes); a radiance sample with a NaN or an infinite component is dropped before RgbFilm::add_sample (the two TODOs of
     * integrator.rs:377-382); safe_acos is acos (math.rs:272-274 calls asin): Sphere (u, v) and SphericalMapping (which then also uses phi
     * for t, texture.rs:972); uniform_hemisphere_pdf = 1/(2 pi) (sampling.rs:306-308 returns 1/(4 pi)); Sphere::pdf_with_context uses
     * 2 pi and multiplies by the squared distance inside the sphere (sphere.rs:438-440, 456); triangle emitters are sampled as PBRT-v4 samples them
     * (round 6): sample_spherical_triangle's barycentrics divided by s1 . e1 (sampling.rs:477: e1 . e1) and renormalised as in PBRT-v4 (:493-497),
     * Triangle::sample_with_context drawing from the warped u whose density it reports (triangle.rs:639-641 shadows it), Triangle::sample negating
     * the normal of a mesh without normals only with reverse_orientation ^ transform_swaps_handedness (triangle.rs:558-560: always); a non-rectangular
     * bilinear patch is area-sampled with PBRT-v4's edge points in sample() and pdf() (bilinear_patch.rs:549-553, 627-628); a shadow ray enters an instance
     * through apply_ray_inverse with its t_max (primitive.rs:173-176 maps it forward) and an instanced hit's interaction is mapped back by PBRT-v4's
     * Transform::operator() (transform.rs:573-608 uses the inverse for everything but the point): instances render as their baked copies. Reference-exact,
     * BASELINE's C1 scene is 19 % darker than an estimator that samples no lights; with 1 it agrees with it (tests/test_quirks_switch.py). */
    uint8_t disable_reference_quirks;
    uint8_t pad[7];
} ShmRenderParams;
enum {
    SHM_INTEGRATOR_PATH = 0,        /* PathIntegrator,       integrator.rs:748-963 */
    SHM_INTEGRATOR_SIMPLE_PATH = 1, /* SimplePathIntegrator, integrator.rs:573-733 */
    SHM_INTEGRATOR_RANDOM_WALK = 2  /* RandomWalkIntegrator, integrator.rs:445-563: le + f cos Li / (1/4pi), folded innermost-first */
};

/* Tile (tile.rs:5-7): Bounds2i, max exclusive. */
typedef struct ShmTile {
    int32_t x0, y0, x1, y1;
} ShmTile;

/* RgbFilmPixel without the unused splat (film.rs:470-479). */
typedef struct ShmFilmPixel {
    double rgb_sum[3];
    double weight_sum;
} ShmFilmPixel;

typedef struct ShmStats {
    uint64_t paths;
    uint64_t rays_closest;     /* camera + extension rays traced by BvhAggregate::intersect */
    uint64_t rays_any;         /* shadow rays traced by intersect_predicate */
    uint64_t nodes_closest;    /* BVH nodes visited (bounds tests) */
    uint64_t tris_closest;     /* primitive tests */
    uint64_t nodes_any;
    uint64_t tris_any;
    double ms_total;           /* HIP-event time of the whole call on the render stream */
    double ms_trace_closest;   /* summed HIP-event time of K2 launches */
    double ms_trace_any;       /* summed HIP-event time of K3 launches */
    double ms_shade;           /* K1 + K5 + K6 */
    uint32_t launches_closest;
    uint32_t launches_any;
    /* ABI v7 (multi-GPU entry points): */
    double ms_gather;          /* HIP-event time of the film gather on this rank's render stream (RCCL send / recv group or xGMI peer copies) */
    uint64_t gather_bytes;     /* film bytes this rank sent (peers) or received (root) */
} ShmStats;

typedef struct ShmRay {
    float o[3];
    float d[3];
    float t_max;
    float pad;
} ShmRay;

/* ShapeIntersection reduced to what identifies it: primitive (leaf-order index), t_hit and the
 * shape-local hit parameters (triangle: b0,b1,b2; sphere: p_obj.xyz in b0..b2 and phi). */
typedef struct ShmHit {
    int32_t prim;   /* -1: miss */
    float t;
    float b0, b1, b2;
    float phi;
    uint32_t instance;  /* 0: a top-level primitive; else 1 + the leaf-order slot of the SHM_SHAPE_INSTANCE primitive it was found
                           through (prim, t and b* are then in that instance's space) */
    uint32_t pad;
} ShmHit;

typedef struct ShmScene ShmScene;

/* Scene lifetime.  `device` is the HIP device ordinal (one process per GPU: pass LOCAL_RANK). */
SHM_API int shm_scene_create(const ShmSceneDesc* desc, int device, ShmScene** out);
SHM_API void shm_scene_destroy(ShmScene* scene);

/* Film accumulation buffer resident in HBM, pixel_bounds-sized, zero-initialised. */
SHM_API int shm_film_clear(ShmScene* scene);
/* Render one spp-wave [sample_begin, sample_end) (integrator.rs:257-260) of the given tiles into the
 * device film (+=).  Blocking.  stats may be NULL.  The tiles of one call must be pairwise disjoint and inside pixel_bounds
 * (Tile::tile's exclusive ownership, which the reference's unsynchronised film writes rely on, integrator.rs:277-295):
 * overlapping tiles return SHM_ERR_INVALID_ARGUMENT. */
SHM_API int shm_render_wave(ShmScene* scene, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles,
                    int32_t sample_begin, int32_t sample_end, ShmStats* stats);
/* Copy the device film to a caller-allocated pixel_bounds-sized row-major array. */
SHM_API int shm_film_read(ShmScene* scene, ShmFilmPixel* film_out);
/* Device pointer + byte size of the film (for device-side gathers, e.g. RCCL through torch). */
SHM_API int shm_film_device_ptr(ShmScene* scene, void** ptr_out, uint64_t* bytes_out);

/* Whole ImageTileIntegrator::render (integrator.rs:226-322) into the DEVICE film (+=, no clear, no read-back): all
 * spp-waves over the given tiles. Consecutive waves of the reference's schedule are fused into launches of up to 64 spp
 * (identical film sums: a pixel's samples are always added in increasing sample_index). */
SHM_API int shm_render_device(ShmScene* scene, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles,
                              ShmStats* stats);
/* The same with clear + read-back: whole ImageTileIntegrator::render (integrator.rs:226-322): all waves 1,1,2,4,...,64,64,... over the
 * given tiles; `film` += on the host. */
SHM_API int shm_render(ShmScene* scene, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles,
               ShmFilmPixel* film, ShmStats* stats);

/* Bring-up / microbenchmark entries for K2/K3 alone (BvhAggregate::intersect / intersect_predicate,
 * aggregate.rs:71-203).  Host arrays in, host arrays out. */
SHM_API int shm_trace_closest(ShmScene* scene, const ShmRay* rays, uint32_t n, ShmHit* hits_out, ShmStats* stats);
SHM_API int shm_trace_any(ShmScene* scene, const ShmRay* rays, uint32_t n, uint8_t* occluded_out, ShmStats* stats);
/* Same, rays already resident in HBM (device pointers); used by bench.py's traversal roofline leg.
 * `repeat` launches back-to-back; stats->ms_trace_* is the HIP-event total over all of them. */
SHM_API int shm_trace_closest_device(ShmScene* scene, const void* rays_dev, uint32_t n, void* hits_dev, int repeat,
                             ShmStats* stats);
SHM_API int shm_trace_any_device(ShmScene* scene, const void* rays_dev, uint32_t n, void* occluded_dev, int repeat,
                         ShmStats* stats);

/* ---- multi-GPU (ABI v7) -------------------------------------------------------------------------------------------
 * Pixels are independent and tile ownership is exclusive (the reference's one parallel region, integrator.rs:242-304, relies on that
 * for its unsynchronised film writes), so the frame shards by tiles with the scene replicated per GPU and NO collective while
 * rendering; one exchange at the end brings every rank's film rows to the root. Sharding is by blocks of whole tile rows, so a
 * rank's pixels are full-width row ranges of the film: only those rows travel (33 MB per rank of the 265 MB film at 3840x2160 on 8
 * GPUs), straight from one device film into the other, and no sum is needed. */

/* Static interleaved sharding of the row-major tile list Tile::tile emits: blocks of `rows_per_block` tile rows, block b belongs to
 * rank b % world. rows_per_block 0 = the default (about 16 blocks per rank: expensive image regions are spread over all ranks).
 * idx_out (capacity n_tiles) receives this rank's tile indices in increasing order. */
SHM_API int shm_shard_tiles(uint32_t n_tiles, uint32_t tiles_per_row, int32_t rank, int32_t world, int32_t rows_per_block,
                            uint32_t* idx_out, uint32_t* n_out);

/* One process per GPU (the layout of `torchrun` / MPI launches): an RCCL communicator per scene. The host distributes the 128-byte
 * unique id from rank 0 to every rank by whatever control channel it has (MPI_Bcast, a TCP store, a file: bench.py uses a directory of
 * files, shimmer_amd/launch.py), then every rank calls shm_dist_init (collective: ncclCommInitRank on the scene's device, then rank 0's
 * shard plan — tile rows per block — is broadcast, so that per-process environments cannot make block ownership differ between ranks). */
#define SHM_DIST_ID_BYTES 128
SHM_API int shm_dist_unique_id(uint8_t id_out[SHM_DIST_ID_BYTES]);
SHM_API int shm_dist_init(ShmScene* scene, int32_t rank, int32_t world, const uint8_t id[SHM_DIST_ID_BYTES]);
SHM_API int shm_dist_finalize(ShmScene* scene);   /* also done by shm_scene_destroy */
/* Whole ImageTileIntegrator::render of the scene's frame on `world` GPUs (collective): clears the device film, renders all spp-waves
 * of THIS rank's tiles (Tile::tile(pixel_bounds, 8, 8) sharded with shm_shard_tiles), then gathers the film rows of every rank into
 * rank 0's device film with one RCCL group (ncclRecv per block on the root, ncclSend on the owners; each peer -> root transfer rides its
 * own xGMI link). On return rank 0's device film (shm_film_read / shm_film_device_ptr) holds the complete frame, bit-identical to a
 * single-GPU render; the other ranks hold their own rows. stats (may be NULL) is this rank's. Without shm_dist_init it renders the
 * whole frame on this GPU (world = 1). */
SHM_API int shm_render_sharded(ShmScene* scene, const ShmRenderParams* params, ShmStats* stats);
/* Small collectives on the scene's communicator, so that a host needs NO second communication stack beside this library (bench.py runs
 * its barrier and its max-over-ranks clock through these; the only thing the host carries itself is the 128-byte id). Host values in,
 * host values out; they go through a 32 KB device scratch and ncclAllReduce / ncclAllGather on the render stream. Every wait polls the
 * stream and ncclCommGetAsyncError instead of blocking: a peer that died or never arrives becomes SHM_ERR_DEVICE after SHM_DIST_TIMEOUT_S
 * (default 900 s), and the communicator is aborted so that the other ranks' pending operations fail too. Without shm_dist_init (or with
 * a world of 1) they are the identity. */
enum { SHM_REDUCE_SUM = 0, SHM_REDUCE_MAX = 1, SHM_REDUCE_MIN = 2 };
SHM_API int shm_dist_barrier(ShmScene* scene);   /* this rank's streams drained, then one all-reduce: every rank arrived */
SHM_API int shm_dist_allreduce_f64(ShmScene* scene, double* values /* in / out */, uint32_t n /* <= 4096 */, int32_t op /* SHM_REDUCE_* */);
SHM_API int shm_dist_allgather_f64(ShmScene* scene, const double* mine, uint32_t n, double* all_out /* world * n <= 4096, rank-major */);
/* What this process runs on: the communicator as RCCL sees it and the shared objects the two runtimes were mapped from (a host process that
 * imported another ROCm stack before this library would show up here). scene may be NULL (library-wide fields only). */
typedef struct ShmDistInfo {
    int32_t rank, world;             /* as given to shm_dist_init (0, 1 without) */
    int32_t rccl_ranks;              /* ncclCommCount of the communicator (0 without one) */
    int32_t rccl_device;             /* ncclCommCuDevice */
    int32_t rccl_version;            /* ncclGetVersion: e.g. 22707 */
    int32_t hip_runtime_version;     /* hipRuntimeGetVersion */
    int32_t rows_per_block;          /* tile rows per shard block in use (rank 0's, broadcast at shm_dist_init) */
    uint32_t n_my_tiles;             /* tiles this rank owns */
    char librccl_path[256];          /* dladdr of ncclGetVersion */
    char libamdhip_path[256];        /* dladdr of hipRuntimeGetVersion */
} ShmDistInfo;
SHM_API int shm_dist_info(ShmScene* scene, ShmDistInfo* out);
SHM_API int shm_device_synchronize(int32_t device);   /* hipDeviceSynchronize on that ordinal */
/* Test entry: sends this rank's film rows to ITSELF through the same RCCL send / recv group into a scratch film and compares
 * (exercises the RCCL transport on a one-GPU box). SHM_OK when every byte matches. */
SHM_API int shm_dist_selftest(ShmScene* scene);

/* One process, n devices: the same frame with one host thread and one scene replica per entry of `devices` (HIP ordinals; an ordinal may
 * repeat, which shares that GPU between two replicas — used by the tests on a one-GPU box), tiles sharded as above, film rows gathered
 * into the first device's film with hipMemcpyPeerAsync over xGMI, then read back. film_out: pixel_bounds-sized, overwritten.
 * stats_per_device: n_devices entries or NULL. A C-only caller renders on every GPU of a node with this one call. */
SHM_API int shm_render_multi(const ShmSceneDesc* desc, const int32_t* devices, int32_t n_devices, const ShmRenderParams* params,
                             ShmFilmPixel* film_out, ShmStats* stats_per_device);

SHM_API const char* shm_last_error(void);   /* thread-local, never NULL */
SHM_API int shm_device_count(void);

/* ---- host-side mirror of the reference's scene-construction steps (no GPU needed) ------------- */

/* BvhAggregate::new (aggregate.rs:207-467): recursive build (split_method 0 = middle, 1 = equal counts),
 * 1-primitive leaves, DFS flatten.  prim_bounds: 6 floats (min xyz, max xyz) per input primitive.
 * nodes_out capacity 2*n-1; prim_order_out[n] receives, per leaf-order slot, the input primitive index. */
SHM_API int shm_bvh_build(const float* prim_bounds, uint32_t n, int split_method, ShmBvhNode* nodes_out,
                  uint32_t* n_nodes_out, uint32_t* prim_order_out);
/* Test entry: the Bounds3f operations the builder is made of (union, union_point, surface_area, volume, max_dimension; bounding_box.rs:394-446),
 * for the reference's own vectors (bounding_box.rs:699-733, 950-995). a, b: {min xyz, max xyz}. out[16]: union(a, b) min, max | union_point(a, p)
 * min, max | surface_area(a), volume(a), max_dimension(a), 0. */
SHM_API int shm_bounds3_probe(const float a[6], const float b[6], const float p[3], float out[16]);
/* (The DEVICE test entry of rounds 4-5, shm_debug_eval_leaf, lives in the test library libshimmer_hip_probe.so since round 6: include/shimmer_hip_probe.h.) */
/* Tile::tile (tile.rs:21-104). tiles_out capacity ceil(w/tw)*ceil(h/th); returns the count via n_out. */
SHM_API int shm_tile_bounds(const int32_t pixel_bounds[4], int32_t tile_w, int32_t tile_h, ShmTile* tiles_out,
                    uint32_t* n_out);
/* PerspectiveCamera::new + CameraTransform (camera.rs:893-963, 507-523, 594-642) for a world_from_camera
 * matrix, fov (degrees), screen window derived from the aspect ratio (camera.rs:848-864). Rendering
 * coordinate system: CameraWorld (the reference default, options.rs). */
SHM_API int shm_camera_perspective(const float world_from_camera[16], float fov_deg, const int32_t full_resolution[2],
                           float lens_radius, float focal_distance, ShmCamera* out,
                           float render_from_world_out[16]);
/* OrthographicCamera::new (camera.rs:713-744) with Transform::orthographic(0, 1) and the same screen window / CameraWorld
 * conventions. lens_radius / focal_distance are stored but unused, as in the reference. */
SHM_API int shm_camera_orthographic(const float world_from_camera[16], const int32_t full_resolution[2], float lens_radius,
                            float focal_distance, ShmCamera* out, float render_from_world_out[16]);
/* The same two constructors with the rest of what Camera::create reads (camera.rs:676-705, 848-890) and the rendering space of
 * CameraTransform::new (camera.rs:507-523; Option "rendercoordsys"): */
enum { SHM_RENDER_SPACE_CAMERA_WORLD = 0, SHM_RENDER_SPACE_CAMERA = 1, SHM_RENDER_SPACE_WORLD = 2 };
typedef struct ShmCameraParams {
    uint32_t kind;                /* SHM_CAMERA_PERSPECTIVE / SHM_CAMERA_ORTHOGRAPHIC */
    uint32_t render_space;        /* SHM_RENDER_SPACE_* */
    float world_from_camera[16];  /* row-major */
    float fov_deg;                /* perspective only */
    int32_t full_resolution[2];
    float lens_radius, focal_distance;
    float frame_aspect_ratio;     /* "frameaspectratio"; 0 = x / y of the film */
    uint32_t has_screen_window;
    float screen_window[4];       /* "screenwindow" as the file gives it: x0 x1 y0 y1 */
} ShmCameraParams;
SHM_API int shm_camera_create(const ShmCameraParams* params, ShmCamera* out, float render_from_world_out[16]);
/* C entry to the C++ host mirror of the reference's integrator interface (shimmer_amd/csrc/host/integrator.hpp):
 * create_integrator(name, {maxdepth, regularize, lightsampler "uniform", spp}, scene)->render(options), integrator.rs:16-42,
 * 52-54, 120-210, 226-322. `name`: "path", "simplepath" (sample_lights / sample_bsdf are its "samplelights" / "samplebsdf") or
 * "randomwalk";
 * anything else fails the way the reference panics ("Unknown integrator ..."); the message
 * is available through shm_last_error(). film_out: pixel_bounds-sized, overwritten; n_waves_out: spp-waves rendered. */
SHM_API int shm_integrator_render(const char* name, const ShmSceneDesc* scene, int device, int32_t max_depth, int regularize,
                          int sample_lights, int sample_bsdf, int32_t samples_per_pixel, int32_t seed, int disable_pixel_jitter,
                          int disable_wavelength_jitter, ShmFilmPixel* film_out, ShmStats* stats_out, int32_t* n_waves_out);
/* RgbFilm::get_image / get_pixel_rgb (film.rs:647-707, 720-738) over a read-back film: rgb = sum / weight_sum (when the
 * weight is non-zero), plus the (here always zero) splat term, times output_rgb_from_sensor_rgb (film.rs:524; row-major 3x3,
 * passed by the caller, who owns the colour space), with the reference's f16 clamp when write_fp16 is set (film.rs:676-690,
 * including its `rgb.g > max -> rgb.r = max` assignment). rgb_out: 3 floats per pixel, same order as `film`. */
SHM_API int shm_film_get_image(const ShmFilmPixel* film, uint64_t n_pixels, const float output_rgb_from_sensor_rgb[9], int write_fp16,
                       float* rgb_out);
/* Image::write_pfm (image.rs:1333-1377): RGB float, bottom-up rows, little-endian (scale -1). */
SHM_API int shm_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height);
/* TriQuadMesh::read_ply (shape/mesh.rs:179-358; the "plymesh" shape, shape/shape.rs:97-135): vertex element with float x y z
 * [nx ny nz] [u v | s t | texture_u texture_v | texture_s texture_t], face element with an int list vertex_indices /
 * vertex_index (triangles and quads; a quad v0 v1 v2 v3 is stored in bilinear-patch order v0 v1 v3 v2) and optionally
 * face_indices. ascii, binary_little_endian and binary_big_endian. n and uv are ALWAYS n_vertices long, zero where the file has
 * no such property — the reference builds them the same way and hands both to the meshes (shape/shape.rs:104-131). Anything
 * the reference panics on (other elements, non-float vertex properties, unsigned index lists, faces that are neither triangles
 * nor quads, indices out of range) returns SHM_ERR_INVALID_ARGUMENT with the message in shm_last_error(). Free with shm_ply_free. */
typedef struct ShmPlyMesh {
    uint32_t n_vertices, n_tri_indices, n_quad_indices, n_face_indices;
    float* p;               /* 3 * n_vertices */
    float* n;               /* 3 * n_vertices */
    float* uv;              /* 2 * n_vertices */
    int32_t* tri_indices;   /* n_tri_indices  (3 per triangle) */
    int32_t* quad_indices;  /* n_quad_indices (4 per quad, patch order) */
    int32_t* face_indices;  /* n_face_indices */
} ShmPlyMesh;
SHM_API int shm_ply_read(const char* filename, ShmPlyMesh* out);
SHM_API void shm_ply_free(ShmPlyMesh* mesh);

/* ---- PBRT-v4 scene front end (ABI v7) ------------------------------------------------------------------------------
 * The reference's loader (loading/tokenizer.rs, parser.rs:216-351, parser_target.rs:50-184, scene.rs:1221-2033) restated in C++ for the
 * directive set the repository's scenes use: transforms (LookAt Translate Scale Rotate Identity Transform ConcatTransform CoordinateSystem
 * CoordSysTransform ReverseOrientation), Camera (perspective / orthographic, incl. frameaspectratio / screenwindow), Film (rgb, whitebalance),
 * Attribute, Option (incl. rendercoordsys), Sampler (independent), PixelFilter (box),
 * Integrator (path / simplepath / randomwalk), Option, WorldBegin, AttributeBegin / End, Material / MakeNamedMaterial / NamedMaterial
 * (diffuse conductor dielectric thindielectric coateddiffuse coatedconductor mix, "normalmap"), Texture (float / spectrum: constant scale
 * mix directionmix imagemap — with the uv / spherical / cylindrical / planar mappings), AreaLightSource (diffuse), LightSource (point,
 * infinite: uniform or an environment image), Shape (trianglemesh bilinearmesh sphere plymesh), ObjectBegin / ObjectEnd / ObjectInstance,
 * Include — with the reference's parameter names and defaults. Spectra: "float", "spectrum" (lambda / value pairs, a named spectrum or a
 * spectrum file), "blackbody", and "rgb" as RgbAlbedo / RgbUnbounded / RgbIlluminantSpectrum by the slot that reads it (paramdict.rs:605-656)
 * through the sRGB rgb2spec coefficient table — the `.spec` file the reference loads from rgbtospec/srgb.spec (rgb_to_spectra.rs:27-31),
 * looked for in $SHM_RGB2SPEC_SRGB, <scene dir>/rgbtospec/srgb.spec, ./rgbtospec/srgb.spec, then the table tools/gen_rgb2spec.py writes
 * beside this library. Image files are PNG, the one format the reference reads (image.rs:1140-1311), decoded and turned into MIP pyramids
 * exactly as Image::read / MIPMap::create_from_file / Image::generate_pyramid do (host/image_io.hpp). What the reference leaves todo!() or
 * this backend does not take (media, Import, portals, animated transforms, other colour spaces) is SHM_ERR_UNSUPPORTED: nothing is silently
 * rendered as something else. Errors carry file:line in shm_last_error(). The returned description owns every array it points to; free
 * it with shm_pbrt_free. */
typedef struct ShmPbrtScene {
    ShmSceneDesc desc;          /* ready for shm_scene_create */
    ShmRenderParams params;     /* Sampler "pixelsamples" / "seed", Integrator "maxdepth" / "regularize" / "samplelights" / "samplebsdf", Option flags */
    char integrator[32];        /* "path" (default), "simplepath", "randomwalk": the name create_integrator receives */
    char output_filename[256];  /* Film "filename" (default "shimmer.pfm", film.rs:232-243) */
    void* owner;                /* the loader's storage */
    float output_rgb_from_sensor_rgb[9]; /* RgbFilm::new (film.rs:524): rgb_from_xyz of sRGB (from its primaries and the D65 white, colorspace.rs:38-72)
                                   times the sensor's matrix — the identity, or the von Kries white balance of Film "whitebalance" (color.rs:404-416,
                                   the illuminant from DenselySampledSpectrum::d, spectrum.rs:215-262). Row-major; what shm_film_get_image takes. */
} ShmPbrtScene;
SHM_API int shm_scene_load_pbrt(const char* path, ShmPbrtScene** out);
/* The same from a string; base_dir (may be NULL) resolves Include and plymesh file names. */
SHM_API int shm_scene_parse_pbrt(const char* text, const char* base_dir, ShmPbrtScene** out);
SHM_API void shm_pbrt_free(ShmPbrtScene* scene);
/* Test entries (the front end's own tokenizer and parameter-list parser, exposed so that the reference's in-source vectors —
 * loading/tokenizer.rs:126-260, token.rs:218-290, param.rs:214-260, parser.rs:656-870 — can be replayed against them):
 * shm_pbrt_tokenize writes every token as <kind letter><token text as the reference's Token holds it, quotes kept><NUL>; kinds: D a
 * directive name (Token::is_directive), W another bare word, S a quoted string (Token::is_quote), B a bracket, Q an opening quote without
 * its partner (the rest of the input, as tokenizer.rs yields it; the loader proper reports it as an error, as the reference's parser does).
 * shm_pbrt_parse_params parses a parameter list (`"type name" value | [ values ]` ...) and writes it as JSON:
 * [{"type", "name", "floats", "ints", "bools", "strings"}] (ParsedParameter, paramdict.rs). */
SHM_API int shm_pbrt_tokenize(const char* text, char* out, uint64_t capacity, uint32_t* n_tokens);
SHM_API int shm_pbrt_parse_params(const char* text, char* out_json, uint64_t capacity);
/* Two pieces of the front end the scene generators of this repository share with the loader, so that both hand the library bit-identical
 * inputs: DenselySampledSpectrum::new(BlackbodySpectrum::new(T)) at 360..=830 nm (spectra/spectrum.rs:430-489) and the world_from_camera
 * matrix of Transform::look_at (transform.rs:270-303), both in f32 as the reference computes them. */
SHM_API int shm_blackbody_dense(float temperature_kelvin, float out471[471]);
SHM_API int shm_look_at(const float eye[3], const float look[3], const float up[3], float world_from_camera_out[16]);
/* The image side of the front end for hosts that fill ShmSceneDesc themselves: Image::read (PNG; `encoding` "sRGB" / "linear" / "gamma <g>",
 * NULL = "sRGB"; image.rs:1140-1311, color.rs:487-525), then — with build_pyramid — MIPMap::create_from_file's channel selection and
 * Image::generate_pyramid for the wrap mode (mipmap.rs:42-99, image.rs:699-802, 1007-1138). Levels come finest first with texel offsets
 * relative to `texels`; channels: 1 ("Y") or 3 (R, G, B; an alpha plane is dropped — file_channels says what the file held, 4 with
 * build_pyramid meaning the alpha was not all ones). has_color_space: the RGB PNG's sRGB colour space (ImageMetadata::color_space). */
typedef struct ShmLoadedImage {
    uint32_t n_levels, n_channels, file_channels, has_color_space;
    uint64_t n_texel_floats;
    ShmImageLevel* levels;
    float* texels;
} ShmLoadedImage;
SHM_API int shm_image_load_png(const char* path, const char* encoding, uint32_t wrap /* SHM_WRAP_* */, int build_pyramid, ShmLoadedImage* out);
SHM_API void shm_image_free(ShmLoadedImage* image);

#ifdef __cplusplus
}
#endif
#endif /* SHIMMER_HIP_H */
