/* render_gltf.c — the C ABI used the way bevy-strolle's `cornell` / `demo` examples use strolle::Engine
 * (bevy-strolle/examples/cornell.rs, demo.rs): load a scene, add a light, create a camera, tick + render a few frames,
 * write the last one as a binary PPM. Plain C99 on purpose: this file is also the proof that include/strolle_hip.h is a
 * C header.
 *
 *   cc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/render_gltf.c \
 *      -L strolle_amd/csrc -lstrolle_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/strolle_amd/csrc -o render_gltf
 *   ./render_gltf scene.glb out.ppm [width height frames]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "strolle_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int status_ = (call);                                                       \
        if (status_ != ST_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, status_, st_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

/* glam::Mat4::look_at_rh inverted = camera-to-world, column-major (what Bevy's GlobalTransform::compute_matrix gives) */
static void camera_to_world(const float eye[3], const float target[3], float out[16]) {
    float f[3], r[3], u[3], n;
    int i;
    for (i = 0; i < 3; i++) f[i] = target[i] - eye[i];
    n = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (i = 0; i < 3; i++) f[i] /= n;
    r[0] = -f[2]; r[1] = 0.0f; r[2] = f[0];          /* cross(f, +Y) */
    n = sqrtf(r[0] * r[0] + r[2] * r[2]);
    for (i = 0; i < 3; i++) r[i] /= n;
    u[0] = r[1] * f[2] - r[2] * f[1]; u[1] = r[2] * f[0] - r[0] * f[2]; u[2] = r[0] * f[1] - r[1] * f[0];
    memset(out, 0, 16 * sizeof(float));
    for (i = 0; i < 3; i++) { out[i] = r[i]; out[4 + i] = u[i]; out[8 + i] = -f[i]; out[12 + i] = eye[i]; }
    out[15] = 1.0f;
}

/* glam::Mat4::perspective_infinite_reverse_rh (Bevy's PerspectiveProjection), column-major */
static void perspective_infinite_reverse(float fov_y, float aspect, float z_near, float out[16]) {
    const float f = 1.0f / tanf(0.5f * fov_y);
    memset(out, 0, 16 * sizeof(float));
    out[0] = f / aspect; out[5] = f; out[11] = -1.0f; out[14] = z_near;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.gltf|scene.glb out.ppm [width height frames]\n", argv[0]);
        return 2;
    }
    const uint32_t width = argc > 3 ? (uint32_t)atoi(argv[3]) : 640u, height = argc > 4 ? (uint32_t)atoi(argv[4]) : 360u;
    const int frames = argc > 5 ? atoi(argv[5]) : 24;

    StEngine* engine = NULL;
    CHECK(st_engine_create(0, &engine));

    StGltfOptions options;
    memset(&options, 0, sizeof options);
    options.first_handle = 1; options.first_image_handle = 1000;
    options.light_radius = 0.15f;                                                       /* cornell.rs:45-54 */
    StGltfSummary scene;
    CHECK(st_scene_load_gltf(engine, argv[1], &options, &scene));
    fprintf(stderr, "%u meshes, %u triangles, %u materials, %u images (%u dropped, %u primitives skipped), %u lights\n", scene.meshes,
            scene.triangles, scene.materials, scene.images, scene.images_dropped, scene.primitives_skipped, scene.lights);

    StLight light;                                                                      /* the example's own lamp, on top of the file's */
    memset(&light, 0, sizeof light);
    light.kind = ST_LIGHT_POINT;
    light.position[0] = 0.0f; light.position[1] = 1.5f; light.position[2] = 0.5f;
    light.radius = 0.15f; light.range = 35.0f;
    light.color[0] = light.color[1] = light.color[2] = 50.0f / (4.0f * 3.14159265f);   /* cornell.rs:45-54 */
    CHECK(st_light_insert(engine, 1 + scene.lights, &light));
    CHECK(st_sun_update(engine, 0.0f, -1.0f));                                          /* night: cornell.rs:87 */

    StCamera camera;
    memset(&camera, 0, sizeof camera);
    camera.mode = ST_MODE_IMAGE; camera.denoise = 1; camera.width = width; camera.height = height;
    {
        const float eye[3] = {0.0f, 1.0f, 3.2f}, target[3] = {0.0f, 1.0f, 0.0f};          /* cornell.rs:76-78 */
        camera_to_world(eye, target, camera.transform);
        perspective_infinite_reverse(3.14159265f / 4.0f, (float)width / (float)height, 0.1f, camera.projection);
    }
    StHandle cam = 0;
    CHECK(st_camera_create(engine, &camera, &cam));
    CHECK(st_camera_set_output_format(engine, cam, ST_FORMAT_RGBA8_UNORM_SRGB));        /* what a swap chain would hold */

    /* The present path of the Rust facade (rust/strolle-hip/src/present.rs), in C: two device frames and two page-locked
     * host frames alternate; frame N's copy to the host is enqueued behind its composition and runs while frame N+1
     * renders, and what gets presented at tick N+1 is frame N (one frame of latency, no stream synchronisation). */
    const size_t frame_bytes = (size_t)width * height * 4;
    void* frame[2] = {NULL, NULL};
    void* host_frame[2] = {NULL, NULL};
    for (int k = 0; k < 2; k++)
        if (hipMalloc(&frame[k], frame_bytes) != hipSuccess || hipHostMalloc(&host_frame[k], frame_bytes, 0) != hipSuccess) { fprintf(stderr, "allocation failed\n"); return 1; }
    unsigned char* host = (unsigned char*)malloc(frame_bytes);   /* stands in for queue.write_texture's destination */
    if (!host) return 1;
    int presented = 0, waited = 0;
    for (int i = 0; i < frames; i++) {        /* the loop of bevy-strolle's render node: update_camera, tick, render_camera */
        const int k = i & 1;
        CHECK(st_camera_update(engine, cam, &camera));
        CHECK(st_tick(engine, NULL));
        CHECK(st_render_camera(engine, cam, frame[k], NULL));
        CHECK(st_camera_present_copy(engine, cam, frame[k], host_frame[k], frame_bytes, NULL));
        if (i > 0) {                          /* present frame i-1: its copy was enqueued a whole frame ago */
            int ready = 0;
            CHECK(st_camera_present_ready(engine, cam, host_frame[k ^ 1], 0, &ready));
            if (!ready) { waited++; CHECK(st_camera_present_ready(engine, cam, host_frame[k ^ 1], 1, &ready)); }
            memcpy(host, host_frame[k ^ 1], frame_bytes);
            presented++;
        }
    }
    {                                         /* the last frame */
        int ready = 0;
        CHECK(st_camera_present_ready(engine, cam, host_frame[(frames - 1) & 1], 1, &ready));
        memcpy(host, host_frame[(frames - 1) & 1], frame_bytes);
        presented++;
    }
    uint64_t rays = 0;
    CHECK(st_camera_ray_count(engine, cam, &rays, 0));

    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; i++) fwrite(host + 4 * i, 1, 3, f);
    fclose(f);
    fprintf(stderr, "%d frames (%d presented, %d waited for their copy), %llu rays, wrote %s\n", frames, presented, waited, (unsigned long long)rays, argv[2]);

    free(host);
    for (int k = 0; k < 2; k++) { (void)hipFree(frame[k]); (void)hipHostFree(host_frame[k]); }
    CHECK(st_camera_delete(engine, cam));
    st_engine_destroy(engine);
    return 0;
}
