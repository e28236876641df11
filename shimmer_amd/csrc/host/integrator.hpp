// shimmer-hip — C++ host mirror of the reference's integrator interface, above the C ABI of include/shimmer_hip.h.
//
// The reference is compiled (Rust) code and no Rust toolchain exists in this image, so the host side a maintainer would
// write as `impl Integrator for WavefrontPathIntegrator` is written here in C++ with the reference's
// names, argument meaning and error behaviour (paths relative to /root/reference/src):
//   integrator.rs:52-54    trait Integrator { fn render(&mut self, options: &Options); }        -> shimmer::Integrator
//   integrator.rs:16-42    create_integrator(name, parameters, camera, sampler, aggregate, ..)   -> shimmer::create_integrator
//   integrator.rs:180-210  ImageTileIntegrator::create_path_integrator ("maxdepth" 5, "regularize" false, "lightsampler"
//                          "uniform")                                                            -> PathIntegratorParameters
//   integrator.rs:226-322  ImageTileIntegrator::render: spp-waves 1,1,2,4,...,64 over 8x8 tiles -> WavefrontPathIntegrator::render
//   options.rs:15-61       Options (the fields this path reads)                                  -> shimmer::Options
//   tile.rs:21-104         Tile::tile                                                            -> shm_tile_bounds
// Everything device-side happens behind shm_* calls; this file contains no arithmetic of the hot path and no fallback.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/shimmer_hip.h"

namespace shimmer {

// options.rs:15-61, the fields the path integrator reads (the others drive the front end, which stays in the reference)
struct Options {
    int32_t seed = 0;
    bool disable_pixel_jitter = false;
    bool disable_wavelength_jitter = false;
    bool force_diffuse = false;
    bool disable_texture_filtering = false;
    bool wavefront = true;  // main.rs:152-155: the flag that selects this backend
};

// what create_path_integrator reads from its ParameterDictionary (integrator.rs:188-192) and from the sampler prototype
// (sampler.rs:95-99)
struct PathIntegratorParameters {
    int32_t max_depth = 5;                  // "maxdepth"
    bool regularize = false;                // "regularize"
    std::string light_sampler = "uniform";  // "lightsampler" (light_sampler.rs:24-40: only "uniform" exists on this path)
    int32_t samples_per_pixel = 16;         // Sampler::samples_per_pixel()
    bool sample_lights = true;              // SimplePath "samplelights" (integrator.rs:135-137)
    bool sample_bsdf = true;                // SimplePath "samplebsdf"
};

// integrator.rs:52-54
class Integrator {
public:
    virtual ~Integrator() = default;
    virtual void render(const Options& options) = 0;
};

// The reference panics; across a C++ boundary that is an exception with the same message.
class IntegratorError : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// ImageTileIntegrator with RayPathLiEvaluator::Path (integrator.rs:149-175, 180-210), on the GPU.
class WavefrontPathIntegrator : public Integrator {
public:
    // `scene` is what the reference hands to create_integrator — camera (+ film), aggregate, lights — already flattened
    // into the ABI's POD description; the arrays are only borrowed for the duration of the constructor.
    WavefrontPathIntegrator(const ShmSceneDesc& scene, const PathIntegratorParameters& parameters, int device = 0,
                            uint8_t integrator = SHM_INTEGRATOR_PATH)
        : params_(parameters), integrator_(integrator) {
        if (parameters.light_sampler != "uniform") throw IntegratorError("Unknown light sampler " + parameters.light_sampler);
        ShmScene* created = nullptr;
        check(shm_scene_create(&scene, device, &created), "shm_scene_create");
        scene_.reset(created);  // owned from here on: a later throw in this constructor still releases the device copies
        const int32_t* pb = scene.film.pixel_bounds;
        width_ = pb[2] - pb[0];
        height_ = pb[3] - pb[1];
        // Tile::tile(pixel_bounds, 8, 8), integrator.rs:235-239
        uint32_t n = 0;
        tiles_.resize((size_t)((width_ + 7) / 8) * (size_t)((height_ + 7) / 8));
        check(shm_tile_bounds(pb, 8, 8, tiles_.data(), &n), "shm_tile_bounds");
        tiles_.resize(n);
        film_.assign((size_t)width_ * (size_t)height_, ShmFilmPixel{});
    }
    ~WavefrontPathIntegrator() override = default;
    WavefrontPathIntegrator(const WavefrontPathIntegrator&) = delete;
    WavefrontPathIntegrator& operator=(const WavefrontPathIntegrator&) = delete;

    // integrator.rs:226-322. The wave loop is the reference's (wave_start, wave_end, next_wave_size); each wave is one
    // shm_render_wave over all tiles (the rayon par_iter of :242 is the GPU's job). At the end the film sums are read
    // back; writing the image stays with the caller (film.rs:647-707 / shm_film_get_image + shm_write_pfm).
    void render(const Options& options) override {
        ShmRenderParams rp{};
        rp.seed = (uint64_t)(int64_t)options.seed;
        rp.samples_per_pixel = params_.samples_per_pixel;
        rp.max_depth = params_.max_depth;
        rp.regularize = params_.regularize ? 1 : 0;
        rp.force_diffuse = options.force_diffuse ? 1 : 0;
        rp.disable_texture_filtering = options.disable_texture_filtering ? 1 : 0;
        rp.integrator = integrator_;
        rp.sample_lights = params_.sample_lights ? 1 : 0;
        rp.sample_bsdf = params_.sample_bsdf ? 1 : 0;
        rp.disable_pixel_jitter = options.disable_pixel_jitter ? 1 : 0;
        rp.disable_wavelength_jitter = options.disable_wavelength_jitter ? 1 : 0;
        stats_ = ShmStats{};
        check(shm_film_clear(scene_.get()), "shm_film_clear");
        const int32_t spp = params_.samples_per_pixel;
        int32_t wave_start = 0, wave_end = 1, next_wave_size = 1;
        waves_ = 0;
        while (wave_start < spp) {
            check(shm_render_wave(scene_.get(), &rp, tiles_.data(), (uint32_t)tiles_.size(), wave_start, wave_end, &stats_), "shm_render_wave");
            ++waves_;
            wave_start = wave_end;
            wave_end = std::min(spp, wave_end + next_wave_size);
            next_wave_size = std::min(2 * next_wave_size, 64);
        }
        check(shm_film_read(scene_.get(), film_.data()), "shm_film_read");
    }

    const std::vector<ShmFilmPixel>& film() const { return film_; }  // RgbFilm pixels {rgb_sum, weight_sum}, row-major
    const ShmStats& stats() const { return stats_; }
    int32_t width() const { return width_; }
    int32_t height() const { return height_; }
    int32_t waves() const { return waves_; }

private:
    static void check(int rc, const char* what) {
        if (rc != SHM_OK) {
            const char* msg = shm_last_error();
            throw IntegratorError(std::string(what) + " failed (" + std::to_string(rc) + "): " + (msg ? msg : ""));
        }
    }
    PathIntegratorParameters params_;
    uint8_t integrator_ = SHM_INTEGRATOR_PATH;
    struct SceneDeleter { void operator()(ShmScene* s) const { shm_scene_destroy(s); } };
    std::unique_ptr<ShmScene, SceneDeleter> scene_;
    std::vector<ShmTile> tiles_;
    std::vector<ShmFilmPixel> film_;
    ShmStats stats_{};
    int32_t width_ = 0, height_ = 0, waves_ = 0;
};

// integrator.rs:16-42: all three names of the reference; anything else is its "Unknown integrator" panic.
inline std::unique_ptr<Integrator> create_integrator(const std::string& name, const PathIntegratorParameters& parameters,
                                                     const ShmSceneDesc& scene, int device = 0) {
    if (name == "path") return std::make_unique<WavefrontPathIntegrator>(scene, parameters, device);
    // ImageTileIntegrator::create_simple_path_integrator (integrator.rs:120-147): the same tile / wave driver around SimplePathIntegrator::li
    if (name == "simplepath") return std::make_unique<WavefrontPathIntegrator>(scene, parameters, device, (uint8_t)SHM_INTEGRATOR_SIMPLE_PATH);
    // ImageTileIntegrator::create_random_walk_integrator (integrator.rs:149-175)
    if (name == "randomwalk") return std::make_unique<WavefrontPathIntegrator>(scene, parameters, device, (uint8_t)SHM_INTEGRATOR_RANDOM_WALK);
    throw IntegratorError("Unknown integrator " + name);
}

}  // namespace shimmer
