// fray_render_mgpu -- the multi-GPU frame over the C ABI from plain C++, one process per GPU:
//
//   fray_render_mgpu scene.fray out.bmp width height spp N
//
// starts N processes (before anything touches the GPU); rank r drives GPU r, renders the 48x48 buckets b with
// b % N == r into its own device frame, and frayhip_gather_buckets moves every rank's buckets into rank 0's frame
// (pack -> grouped RCCL send/recv peer -> root over xGMI -> unpack).  Rank 0 writes the picture.  The communicator
// id travels from rank 0 to the others through a file -- any channel the host has will do (MPI, a socket).
#define __HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "frayhip.h"

static int fail(int rank, const char* what)
{
    fprintf(stderr, "rank %d: %s: %s\n", rank, what, frayhip_last_error());
    return 1;
}

static int rank_main(int rank, int world, int argc, char** argv, const std::string& idFile)
{
    frayhip_host_scene* hs = nullptr;
    if (frayhip_scene_parse(argv[1], &hs) != FRAYHIP_OK) return fail(rank, "Could not parse the scene");
    frayhip_scene_desc* d = frayhip_host_scene_desc(hs);
    d->settings.frameWidth = atoi(argv[3]);
    d->settings.frameHeight = atoi(argv[4]);
    if (d->settings.gi) d->settings.numPaths = atoi(argv[5]);
    else if (d->camera.dof) d->camera.numDOFSamples = atoi(argv[5]);
    d->settings.interactive = 0;
    // FRAY_RENDER_MGPU_ONE_DEVICE: rehearsal on a one-GPU box (every rank on GPU 0, if the RCCL build admits that)
    if (frayhip_init(getenv("FRAY_RENDER_MGPU_ONE_DEVICE") ? 0 : rank) != FRAYHIP_OK) return fail(rank, "Cannot set up the GPU");
    // communicator: rank 0 makes the id, the others wait for the file
    unsigned char id[FRAYHIP_COMM_ID_BYTES];
    if (world > 1) {
        if (rank == 0) {
            if (frayhip_comm_unique_id(id) != FRAYHIP_OK) return fail(rank, "No RCCL");
            FILE* f = fopen((idFile + ".tmp").c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return fail(rank, "Cannot write the id file");
            fclose(f);
            rename((idFile + ".tmp").c_str(), idFile.c_str());
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 3000 && !(f = fopen(idFile.c_str(), "rb")); tries++) usleep(10000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) return fail(rank, "No id from rank 0");
            fclose(f);
        }
    }
    frayhip_comm* comm = nullptr;
    if (frayhip_comm_create(id, rank, world, &comm) != FRAYHIP_OK) return fail(rank, "Cannot create the communicator");
    frayhip_scene* scene = nullptr;
    if (frayhip_scene_create(d, &scene) != FRAYHIP_OK) return fail(rank, "Cannot upload the scene");
    const int W = d->settings.frameWidth, H = d->settings.frameHeight;
    float* d_frame = nullptr;
    if (hipMalloc((void**)&d_frame, (size_t)W * H * 12) != hipSuccess || hipMemset(d_frame, 0, (size_t)W * H * 12) != hipSuccess) return fail(rank, "hipMalloc");
    frayhip_frame f = {FRAYHIP_MODE_RENDER, 42u, rank, world, 0, 0};
    auto t0 = std::chrono::steady_clock::now();
    if (frayhip_render_device(scene, &f, d_frame, nullptr, nullptr, nullptr, nullptr) != FRAYHIP_OK) return fail(rank, "Render failed");
    if (frayhip_gather_buckets(comm, d_frame, W, H, 3, 0, nullptr) != FRAYHIP_OK) return fail(rank, "Gather failed");
    if (hipDeviceSynchronize() != hipSuccess) return fail(rank, "hipDeviceSynchronize");
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) {
        std::vector<float> vfb((size_t)W * H * 3);
        if (hipMemcpy(vfb.data(), d_frame, vfb.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(rank, "hipMemcpy");
        printf("Render took %.2fs on %d GPU%s\n", sec, world, world > 1 ? "s" : "");
        if (frayhip_save_bmp(argv[2], vfb.data(), W, H) != FRAYHIP_OK) return fail(rank, "Cannot write the image");
    }
    (void)hipFree(d_frame);
    frayhip_comm_destroy(comm);
    frayhip_scene_destroy(scene);
    frayhip_host_scene_free(hs);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s scene.fray out.bmp width height spp N\n", argv[0]); return 2; }
    const int world = atoi(argv[6]);
    if (world < 1 || world > 64) { fprintf(stderr, "bad N\n"); return 2; }
    const std::string idFile = "/tmp/fray_render_mgpu_id_" + std::to_string((long)getpid());
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {          // forked before this process has made a single HIP call
        pid_t p = fork();
        if (p == 0) _exit(rank_main(r, world, argc, argv, idFile));
        if (p < 0) { perror("fork"); return 1; }
        kids.push_back(p);
    }
    int rc = rank_main(0, world, argc, argv, idFile);
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = rc ? rc : 1;
    }
    unlink(idFile.c_str());
    if (!rc) printf("Exited cleanly\n");
    return rc;
}
