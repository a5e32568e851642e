// probe: do kernels of two PROCESSES run side by side on one GPU?  Each of the two processes launches a kernel that raises its own
// flag in a buffer both map (HIP IPC) and spins (bounded) until it sees the other's -- the hand-off a device-driven peer-to-peer
// exchange between ranks on one device would rely on.  usage: ipc_spin_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s at %d\n", (int)getpid(), hipGetErrorString(e_), __LINE__); _exit(1); } } while (0)
struct Ctl { hipIpcMemHandle_t h; volatile int ready, opened; };
__global__ void k_meet(unsigned long long* flags, int me, long long* waited) {
    __hip_atomic_store(&flags[me], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    long long n = 0;
    while (__hip_atomic_load(&flags[1 - me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull && n < (1ll << 24)) { __builtin_amdgcn_s_sleep(8); ++n; }
    waited[0] = wall_clock64() - t0; waited[1] = n;
}
int main() {
    Ctl* c = (Ctl*)mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset(c, 0, sizeof *c);
    const pid_t pid = fork();
    const int me = pid == 0 ? 1 : 0;
    unsigned long long* flags = nullptr;
    if (me == 0) {
        CK(hipMalloc(&flags, 256)); CK(hipMemset(flags, 0, 256));
        CK(hipIpcGetMemHandle(&c->h, flags));
        c->ready = 1;
        while (!c->opened) usleep(100);
    } else {
        while (!c->ready) usleep(100);
        CK(hipIpcOpenMemHandle((void**)&flags, c->h, hipIpcMemLazyEnablePeerAccess));
        c->opened = 1;
    }
    long long* w; CK(hipMalloc(&w, 16));
    if (me == 1) usleep(200000);                       // the second process comes 200 ms late: the first one's kernel must still be waiting
    hipLaunchKernelGGL(k_meet, dim3(1), dim3(1), 0, 0, flags, me, w);
    CK(hipDeviceSynchronize());
    long long h[2]; CK(hipMemcpy(h, w, 16, hipMemcpyDeviceToHost));
    printf("[process %d] saw the other's flag after %lld polls, %.1f ms %s\n", me, h[1], h[0] / 1e5, h[1] >= (1ll << 24) ? "(TIMED OUT)" : "");
    if (me == 1) _exit(0);
    int st; wait(&st);
    return 0;
}
