/* The drop-in boundary is a C ABI: this translation unit is compiled as C99 against include/imcui_hip.h
 * (no C++, no torch) and binds the library the way a non-Python host would.  Run by tests/test_host_cpu.py
 * on the CPU box: only entry points that need no GPU are called. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "imcui_hip.h"

typedef int (*fn_version)(void);
typedef size_t (*fn_size)(void);
typedef int (*fn_count)(void);
typedef const char* (*fn_name)(int);
typedef float (*fn_pack)(const float*, int, int, unsigned short*, unsigned short*);

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    void* so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!so) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    fn_version version;
    fn_size sp_floats, lg_floats;
    fn_count lg_tensors;
    fn_name lg_name;
    fn_pack pack;
    *(void**)(&version) = dlsym(so, "imcui_hip_version");
    *(void**)(&sp_floats) = dlsym(so, "imcui_hip_superpoint_packed_floats");
    *(void**)(&lg_floats) = dlsym(so, "imcui_hip_lightglue_packed_floats");
    *(void**)(&lg_tensors) = dlsym(so, "imcui_hip_lightglue_num_tensors");
    *(void**)(&lg_name) = dlsym(so, "imcui_hip_lightglue_tensor_name");
    *(void**)(&pack) = dlsym(so, "imcui_hip_linear_pack_split");
    if (!version || !sp_floats || !lg_floats || !lg_tensors || !lg_name || !pack) return 3;
    /* a host-side packer: 32 x 16 weight -> one fragment per plane */
    float w[32 * 16];
    unsigned short hi[32 * 16], lo[32 * 16];
    int i;
    for (i = 0; i < 32 * 16; ++i) w[i] = 0.01f * (float)(i % 37) - 0.15f;
    {
        const float inv = pack(w, 32, 16, hi, lo);
        printf("version=%d sp_floats=%lu lg_floats=%lu lg_tensors=%d first=%s pack_scale=%g\n", version(), (unsigned long)sp_floats(),
               (unsigned long)lg_floats(), lg_tensors(), lg_name(0), (double)inv);
        if (!(inv > 0.0f) || strcmp(lg_name(0), "posenc.Wr.weight") != 0) return 4;
    }
    dlclose(so);
    return 0;
}
