/* dist_tiles.c — the multi-GPU path of the C ABI (st_dist_*, include/strolle_hip.h) from plain C99: one frame split into tiles,
 * every tile rendered by its own engine, the tiles gathered into rank 0's frame — and checked, byte for byte, against the same
 * frame rendered by ONE engine (CameraMode::Reference pixels read nothing but their own, so the partition must be invisible).
 *
 * So that it runs on a box with one GPU, the `world` engines live in THIS process on device 0 and hand their tiles over through the
 * in-process transport (st_dist_init_local). A real deployment is one process per GPU, and differs in exactly the two marked places:
 *   (1) rank 0 calls st_dist_unique_id() and sends the 128 bytes to the other processes its own way (a socket, a pipe, MPI);
 *   (2) every process calls st_dist_init(engine, rank, world, &id) instead of st_dist_init_local.
 * Everything else — st_dist_set_partition, render, st_dist_gather behind every frame, st_dist_wait before rank 0 presents — is the
 * same code, and the gather then travels as grouped ncclSend / ncclRecv over RCCL.
 *
 *   cc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/dist_tiles.c \
 *      -L strolle_amd/csrc -lstrolle_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/strolle_amd/csrc -o dist_tiles
 *   ./dist_tiles scene.glb world [width height frames]
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

#define MAX_WORLD 16

static void camera_to_world(const float eye[3], const float target[3], float out[16]) {   /* as in render_gltf.c */
    float f[3], r[3], u[3], n;
    int i;
    for (i = 0; i < 3; i++) f[i] = target[i] - eye[i];
    n = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (i = 0; i < 3; i++) f[i] /= n;
    r[0] = -f[2]; r[1] = 0.0f; r[2] = f[0];
    n = sqrtf(r[0] * r[0] + r[2] * r[2]);
    for (i = 0; i < 3; i++) r[i] /= n;
    u[0] = r[1] * f[2] - r[2] * f[1]; u[1] = r[2] * f[0] - r[0] * f[2]; u[2] = r[0] * f[1] - r[1] * f[0];
    memset(out, 0, 16 * sizeof(float));
    for (i = 0; i < 3; i++) { out[i] = r[i]; out[4 + i] = u[i]; out[8 + i] = -f[i]; out[12 + i] = eye[i]; }
    out[15] = 1.0f;
}
static void perspective_infinite_reverse(float fov_y, float aspect, float z_near, float out[16]) {
    const float f = 1.0f / tanf(0.5f * fov_y);
    memset(out, 0, 16 * sizeof(float));
    out[0] = f / aspect; out[5] = f; out[11] = -1.0f; out[14] = z_near;
}

/* one engine with the scene, the lamp and the camera: every rank builds the same state (the scene is replicated) */
static int make_engine(const char* scene_path, const StCamera* camera, StEngine** engine, StHandle* cam) {
    StGltfOptions options;
    StGltfSummary scene;
    StLight light;
    CHECK(st_engine_create(0, engine));
    memset(&options, 0, sizeof options);
    options.first_handle = 1; options.first_image_handle = 1000; options.light_radius = 0.15f;
    CHECK(st_scene_load_gltf(*engine, scene_path, &options, &scene));
    memset(&light, 0, sizeof light);
    light.kind = ST_LIGHT_POINT;
    light.position[0] = 0.0f; light.position[1] = 1.5f; light.position[2] = 0.5f;
    light.radius = 0.15f; light.range = 35.0f;
    light.color[0] = light.color[1] = light.color[2] = 50.0f / (4.0f * 3.14159265f);
    CHECK(st_light_insert(*engine, 1 + scene.lights, &light));
    CHECK(st_sun_update(*engine, 0.0f, -1.0f));
    CHECK(st_set_seed(*engine, 7));           /* the same seed everywhere: the noise is keyed on absolute pixel coordinates */
    CHECK(st_camera_create(*engine, camera, cam));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.gltf|scene.glb world [width height frames]\n", argv[0]);
        return 2;
    }
    const int world = atoi(argv[2]);
    const uint32_t width = argc > 3 ? (uint32_t)atoi(argv[3]) : 640u, height = argc > 4 ? (uint32_t)atoi(argv[4]) : 360u;
    const int frames = argc > 5 ? atoi(argv[5]) : 3;
    if (world < 1 || world > MAX_WORLD) { fprintf(stderr, "world must be 1..%d\n", MAX_WORLD); return 2; }

    StCamera camera;
    memset(&camera, 0, sizeof camera);
    camera.mode = ST_MODE_REFERENCE; camera.depth = 1; camera.width = width; camera.height = height;
    {
        const float eye[3] = {0.0f, 1.0f, 3.2f}, target[3] = {0.0f, 1.0f, 0.0f};
        camera_to_world(eye, target, camera.transform);
        perspective_infinite_reverse(3.14159265f / 4.0f, (float)width / (float)height, 0.1f, camera.projection);
    }
    const size_t frame_bytes = (size_t)width * height * 16;   /* RGBA32F, the default output format */

    /* ---- the frame as ONE engine renders it */
    StEngine* single = NULL;
    StHandle single_cam = 0;
    void* single_frame = NULL;
    if (make_engine(argv[1], &camera, &single, &single_cam)) return 1;
    if (hipMalloc(&single_frame, frame_bytes) != hipSuccess) { fprintf(stderr, "allocation failed\n"); return 1; }
    for (int i = 0; i < frames; i++) {
        CHECK(st_camera_update(single, single_cam, &camera));
        CHECK(st_tick(single, NULL));
        CHECK(st_render_camera(single, single_cam, single_frame, NULL));
    }

    /* ---- the same frame as `world` ranks render it */
    StEngine* engine[MAX_WORLD];
    StHandle cam[MAX_WORLD];
    void* frame[MAX_WORLD][2];          /* every rank alternates two render targets: frame i's gather runs under frame i + 1 */
    void* full[2] = {NULL, NULL};       /* rank 0's assembled frames */
    const uint64_t group = 1;           /* the local transport's mailbox (any number; one per set of cooperating engines) */
    for (int r = 0; r < world; r++) {
        StDistRect owned, window;
        if (make_engine(argv[1], &camera, &engine[r], &cam[r])) return 1;
        /* (1) + (2): one process per GPU would exchange st_dist_unique_id()'s bytes and call st_dist_init(engine, r, world, &id) here */
        CHECK(st_dist_init_local(engine[r], r, world, group));
        CHECK(st_dist_set_partition(engine[r], cam[r], 0 /* default grid: 2 bands, 2 x 2, 4 x 2 */, 0 /* apron: Reference needs none */, &owned, &window));
        fprintf(stderr, "rank %d: tile x %u..%u y %u..%u\n", r, owned.x0, owned.x1, owned.y0, owned.y1);
        for (int k = 0; k < 2; k++) if (hipMalloc(&frame[r][k], frame_bytes) != hipSuccess) { fprintf(stderr, "allocation failed\n"); return 1; }
    }
    for (int k = 0; k < 2; k++) if (hipMalloc(&full[k], frame_bytes) != hipSuccess || hipMemset(full[k], 0, frame_bytes) != hipSuccess) { fprintf(stderr, "allocation failed\n"); return 1; }
    for (int i = 0; i < frames; i++) {
        const int k = i & 1;
        /* with the in-process transport the peers hand their tiles over BEFORE the root collects them: ranks world-1 .. 0 */
        for (int r = world - 1; r >= 0; r--) {
            CHECK(st_camera_update(engine[r], cam[r], &camera));
            CHECK(st_tick(engine[r], NULL));
            CHECK(st_render_camera(engine[r], cam[r], frame[r][k], NULL));     /* only the rank's window is written */
            CHECK(st_dist_gather(engine[r], cam[r], frame[r][k], r == 0 ? full[k] : NULL, NULL));   /* returns at once */
        }
    }
    CHECK(st_dist_wait(engine[0], cam[0], NULL, NULL, 1));      /* before rank 0 hands the last frame on */
    float gather_ms = 0.0f;
    CHECK(st_dist_gather_ms(engine[0], cam[0], &gather_ms));

    /* ---- the partition must be invisible */
    unsigned char* a = (unsigned char*)malloc(frame_bytes);
    unsigned char* b = (unsigned char*)malloc(frame_bytes);
    if (!a || !b) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpy(a, single_frame, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b, full[(frames - 1) & 1], frame_bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    size_t differing = 0;
    double sum = 0.0;
    for (size_t i = 0; i < frame_bytes; i++) differing += a[i] != b[i];
    for (size_t i = 0; i < (size_t)width * height * 4; i++) sum += ((const float*)a)[i];
    fprintf(stderr, "%d ranks, %d frames, last gather %.3f ms on the communication stream: %zu of %zu bytes differ from the single-engine frame (sum of its channels %.3f)\n",
            world, frames, gather_ms, differing, frame_bytes, sum);

    free(a); free(b);
    for (int r = 0; r < world; r++) {
        CHECK(st_dist_shutdown(engine[r]));
        for (int k = 0; k < 2; k++) (void)hipFree(frame[r][k]);
        CHECK(st_camera_delete(engine[r], cam[r]));
        st_engine_destroy(engine[r]);
    }
    for (int k = 0; k < 2; k++) (void)hipFree(full[k]);
    (void)hipFree(single_frame);
    CHECK(st_camera_delete(single, single_cam));
    st_engine_destroy(single);
    return differing == 0 && sum > 0.0 ? 0 : 3;
}
