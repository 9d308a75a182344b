/* frt_cabi_example.c — the C ABI (include/frt.h) driven from plain C, with no Python and no C++ in between: what a Rust / C host links against.
 * Mirrors State::new + a few State::update / State::render turns of the reference (src/state.rs:35-80, :146-224): build the Cornell Box through the
 * SceneBuilder calls, create a Renderer (one GPU) and a multi-device renderer (two strips), render the same frames with both, compare the display
 * buffers byte for byte, print one line. Built and run by tests/test_cabi_example.py (gcc, -lfrt; needs a HIP device).
 *   gcc -std=c99 -O1 -I include tests/cabi/frt_cabi_example.c -L fast-raytracing-wgpu_amd/lib -lfrt -Wl,-rpath,... -o tests/cabi/_build/frt_cabi_example */
#include "frt.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, frt_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t W = 320, H = 200, frames = 4;
    if (frt_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", "the library has no CPU path"); return 2; }
    frt_scene* scene = frt_scene_create_cornell_box();                    /* scenes::create_cornell_box, src/scene/scenes.rs:9 */
    if (!scene) { fprintf(stderr, "scene: %s\n", frt_last_error()); return 1; }
    uint32_t counts[8];
    CHECK(frt_scene_counts(scene, counts));

    frt_render_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.max_depth = 8; opts.flags = FRT_FLAG_PIPELINE;
    frt_renderer* one = frt_renderer_create(scene, W, H, &opts);          /* Renderer::new, src/renderer.rs:206 */
    if (!one) { fprintf(stderr, "renderer: %s\n", frt_last_error()); return 1; }
    const int32_t devices[2] = {0, 0};                                    /* two strips on the one GPU of the test box */
    frt_multi_renderer* two = frt_multi_renderer_create(scene, W, H, 2, devices, &opts);
    if (!two) { fprintf(stderr, "multi renderer: %s\n", frt_last_error()); return 1; }

    for (uint32_t f = 0; f < frames; ++f) {
        frt_camera_uniform cam;                                            /* CameraController::build_uniform at the initial pose, src/camera.rs:207 */
        frt_camera_default((float)W / (float)H, frt_renderer_frame_count(one), counts[3], &cam);
        CHECK(frt_renderer_render(one, &cam));                             /* Renderer::render, src/renderer.rs:349 */
        CHECK(frt_multi_renderer_render(two, &cam));
    }
    uint8_t* a = (uint8_t*)malloc((size_t)W * H * 4);
    uint8_t* b = (uint8_t*)malloc((size_t)W * H * 4);
    CHECK(frt_renderer_read_display(one, a));                              /* post_processed_texture, src/state.rs:226-278 */
    CHECK(frt_multi_renderer_read_display(two, b));
    unsigned long long sum = 0;
    for (size_t i = 0; i < (size_t)W * H * 4; ++i) sum += a[i];
    frt_stats s1, s2;
    CHECK(frt_renderer_stats(one, &s1));
    CHECK(frt_multi_renderer_stats(two, &s2));
    const int same = memcmp(a, b, (size_t)W * H * 4) == 0 && s1.rays_closest == s2.rays_closest && s1.rays_any == s2.rays_any;
    printf("{\"frames\": %u, \"tris\": %u, \"display_sum\": %llu, \"rays\": %llu, \"multi_equals_single\": %s}\n",
           frt_renderer_frame_count(one), counts[0], sum, (unsigned long long)(s1.rays_closest + s1.rays_any), same ? "true" : "false");
    free(a); free(b);
    frt_multi_renderer_destroy(two);
    frt_renderer_destroy(one);
    frt_scene_destroy(scene);
    (void)argc; (void)argv;
    return same ? 0 : 3;
}
