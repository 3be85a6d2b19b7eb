/* Plain-C client of the drop-in boundary (include/mi_engine.h): loads libmi_engine.so, lists the tasks' dimensions and lays an engine out
 * on a HOST buffer (no GPU needed for that; the launching entry points refuse a non-device arena).  Shows that the header is C, that
 * the library needs nothing but plain pointers and sizes, and how a non-Python host (the reference has none, but a C++/Go/Rust trainer
 * would) binds it.   gcc -std=c99 -Iinclude examples/c_abi_probe.c -ldl -o c_abi_probe && ./c_abi_probe isaacgymenvs_amd/libmi_engine.so */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi_engine.h"

#define LOAD(name) do { *(void**)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "isaacgymenvs_amd/libmi_engine.so";
    void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen(%s): %s\n", path, dlerror()); return 1; }
    int (*p_mi_abi_version)(void);
    int (*p_mi_task_info)(const char*, MiTaskInfo*);
    size_t (*p_mi_engine_arena_bytes)(const char*, int);
    int (*p_mi_engine_create)(const char*, const MiSimParams*, const void*, size_t, int, int, uint64_t, void*, size_t, MiEngine**);
    int (*p_mi_engine_num_tensors)(const MiEngine*);
    int (*p_mi_engine_tensor_desc)(const MiEngine*, int, MiTensorDesc*);
    int (*p_mi_engine_step)(MiEngine*, const float*, void*);
    void (*p_mi_engine_destroy)(MiEngine*);
    const char* (*p_mi_last_error)(void);
    LOAD(mi_abi_version); LOAD(mi_task_info); LOAD(mi_engine_arena_bytes); LOAD(mi_engine_create); LOAD(mi_engine_num_tensors);
    LOAD(mi_engine_tensor_desc); LOAD(mi_engine_step); LOAD(mi_engine_destroy); LOAD(mi_last_error);
    if (p_mi_abi_version() != MI_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 3; }

    const char* tasks[] = {"Cartpole", "Ant", "Humanoid", "AnymalTerrain", "ShadowHand", "Anymal", "Quadcopter"};
    for (unsigned i = 0; i < sizeof(tasks) / sizeof(tasks[0]); ++i) {
        MiTaskInfo info;
        if (p_mi_task_info(tasks[i], &info) != 0) { fprintf(stderr, "%s: %s\n", tasks[i], p_mi_last_error()); return 4; }
        printf("%-14s obs %3d  actions %2d  dofs %2d  bodies %2d  params %4d B  arena(4096 envs) %.1f MB\n", tasks[i], info.num_obs, info.num_actions,
               info.num_dofs, info.num_bodies, info.task_params_bytes, p_mi_engine_arena_bytes(tasks[i], 4096) / 1048576.0);
    }
    /* lay Ant out on a host buffer and walk the tensor table */
    const int n = 64;
    size_t bytes = p_mi_engine_arena_bytes("Ant", n);
    void* arena = calloc(1, bytes);
    MiSimParams sim;
    memset(&sim, 0, sizeof sim);
    sim.dt = 0.0166f; sim.substeps = 2; sim.iters = 4;
    MiLocoParams lp;
    memset(&lp, 0, sizeof lp);
    MiEngine* e = NULL;
    if (p_mi_engine_create("Ant", &sim, &lp, sizeof lp, n, 0, 42u, arena, bytes, &e) != 0) { fprintf(stderr, "create: %s\n", p_mi_last_error()); return 5; }
    int nt = p_mi_engine_num_tensors(e);
    for (int i = 0; i < nt; ++i) {
        MiTensorDesc d;
        if (p_mi_engine_tensor_desc(e, i, &d) != 0) return 6;
        if (!strcmp(d.name, "root_states") || !strcmp(d.name, "obs_buf") || !strcmp(d.name, "reset_buf"))
            printf("  %-12s dtype %d  shape [%lld, %lld]  stride [%lld, %lld]  offset %lld\n", d.name, d.dtype, (long long)d.shape[0],
                   (long long)d.shape[1], (long long)d.stride[0], (long long)d.stride[1], (long long)d.byte_offset);
    }
    float* actions = (float*)calloc((size_t)n * 8, sizeof(float));
    int rc = p_mi_engine_step(e, actions, NULL);   /* refused: the arena is host memory */
    printf("step on a host arena -> rc %d (%s)\n", rc, p_mi_last_error());
    p_mi_engine_destroy(e);
    free(actions); free(arena);
    dlclose(lib);
    return rc == 0 ? 7 : 0;
}
