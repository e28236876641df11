/* render_pbrt.c — a C-only caller of libshimmer_hip.so: load a PBRT-v4 scene file, render it on every visible GPU, write a PFM.
 * The whole of the reference's `shimmer scene.pbrt` run (main.rs -> parse -> render_cpu -> ImageTileIntegrator::render -> write_image)
 * through the C ABI of include/shimmer_hip.h, with no Python and no torch in the process.
 *
 *   gcc -O2 -I include examples/render_pbrt.c -L shimmer_amd/csrc -lshimmer_hip -Wl,-rpath,$PWD/shimmer_amd/csrc -o render_pbrt
 *   ./render_pbrt scene.pbrt [out.pfm] [--spp N] [--devices K]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "shimmer_hip.h"

static int die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, shm_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s scene.pbrt [out.pfm] [--spp N] [--devices K]\n", argv[0]);
        return 2;
    }
    const char* out_path = NULL;
    int spp = 0, n_devices = 0;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--spp") && i + 1 < argc) spp = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--devices") && i + 1 < argc) n_devices = atoi(argv[++i]);
        else out_path = argv[i];
    }
    ShmPbrtScene* s = NULL;
    if (shm_scene_load_pbrt(argv[1], &s) != SHM_OK) return die("shm_scene_load_pbrt");
    if (spp > 0) s->params.samples_per_pixel = spp;
    if (!out_path) out_path = s->output_filename;
    const int visible = shm_device_count();
    if (visible < 1) { fprintf(stderr, "no HIP device visible (there is no CPU fallback)\n"); return 1; }
    if (n_devices <= 0 || n_devices > visible) n_devices = visible;
    int32_t devices[64];
    for (int i = 0; i < n_devices && i < 64; ++i) devices[i] = i;

    const int32_t* pb = s->desc.film.pixel_bounds;
    const int32_t w = pb[2] - pb[0], h = pb[3] - pb[1];
    ShmFilmPixel* film = (ShmFilmPixel*)calloc((size_t)w * (size_t)h, sizeof(ShmFilmPixel));
    ShmStats* stats = (ShmStats*)calloc((size_t)n_devices, sizeof(ShmStats));
    float* rgb = (float*)malloc(sizeof(float) * 3 * (size_t)w * (size_t)h);
    if (!film || !stats || !rgb) { fprintf(stderr, "out of memory\n"); return 1; }
    /* ImageTileIntegrator::render on every device: tiles sharded inside the library, film rows gathered over xGMI */
    if (shm_render_multi(&s->desc, devices, n_devices, &s->params, film, stats) != SHM_OK) return die("shm_render_multi");
    unsigned long long rays = 0;
    double ms = 0.0;
    for (int i = 0; i < n_devices; ++i) {
        rays += stats[i].rays_closest + stats[i].rays_any;
        if (stats[i].ms_total > ms) ms = stats[i].ms_total;
    }
    /* RgbFilm::get_image with the film's output matrix (XYZ sensor -> linear sRGB, times the white balance if the Film asked for one), then
     * Image::write_pfm */
    if (shm_film_get_image(film, (uint64_t)w * (uint64_t)h, s->output_rgb_from_sensor_rgb, 0, rgb) != SHM_OK) return die("shm_film_get_image");
    if (shm_write_pfm(out_path, rgb, w, h) != SHM_OK) return die("shm_write_pfm");
    printf("%s: %dx%d, %d spp, integrator %s, %d device(s): %llu rays, %.1f ms on the slowest device -> %s\n", argv[1], w, h, s->params.samples_per_pixel,
           s->integrator, n_devices, rays, ms, out_path);
    free(rgb); free(stats); free(film);
    shm_pbrt_free(s);
    return 0;
}
