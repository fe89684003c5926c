// unregister_probe.hip -- what does the HIP runtime still know about a heap block after hipHostUnregister?
// (HISTORY.md section 10: the one GPU page fault of the round-4 soak was a write of the runtime's own pageable D2H copy to
// a HOST heap address; hypothesis: a heap block that was registered, unregistered, freed and handed out again by malloc.)
// This probe only QUERIES the runtime (hipPointerGetAttributes, hipHostGetDevicePointer): no kernel, no copy, nothing
// that could fault.  Build: hipcc -O2 -o unregister_probe unregister_probe.hip
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static void report(const char *when, void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    void *d = nullptr;
    hipError_t e2 = hipHostGetDevicePointer(&d, p, 0);
    printf("%-44s ptr %p: hipPointerGetAttributes -> %s (type %d, isManaged %d, hostPointer %p, devicePointer %p); "
           "hipHostGetDevicePointer -> %s (%p)\n",
           when, p, hipGetErrorName(e), (int)a.type, (int)a.isManaged, a.hostPointer, a.devicePointer, hipGetErrorName(e2), d);
    (void)hipGetLastError();
}

int main()
{
    mallopt(M_MMAP_THRESHOLD, 1 << 30);   // keep the blocks on the brk heap, like a long-running process's allocator does
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    const size_t bytes = 4u << 20;
    hipSetDevice(0);
    void *warm = nullptr;
    hipMalloc(&warm, 1 << 20);
    for (int round = 0; round < 3; ++round) {
        char *p = (char *)malloc(bytes);
        memset(p, 1, bytes);
        printf("---- round %d\n", round);
        report("fresh heap block", p);
        hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
        printf("hipHostRegister -> %s\n", hipGetErrorName(e));
        report("registered", p);
        e = hipHostUnregister(p);
        printf("hipHostUnregister -> %s\n", hipGetErrorName(e));
        report("after hipHostUnregister", p);
        free(p);
        char *q = (char *)malloc(bytes);
        printf("free + malloc: %s address\n", q == p ? "SAME" : "different");
        report("re-allocated block (never registered)", q);
        free(q);
    }
    hipFree(warm);
    return 0;
}
