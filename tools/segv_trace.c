/* Diagnostic aid (test infrastructure, not part of the product): on SIGSEGV / SIGBUS / SIGABRT print the faulting
 * address, the registers and the call chain of the faulting thread as "library+offset", then let the default action run.
 *   gcc -O1 -g -shared -fPIC -o tools/libsegv_trace.so tools/segv_trace.c -ldl
 * Loaded with ctypes.CDLL (the constructor installs the handlers). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

static void put(const char* s) { (void)!write(2, s, strlen(s)); }

static void where(void* pc) {
    Dl_info info;
    char b[512];
    if (dladdr(pc, &info) && info.dli_fname) {
        snprintf(b, sizeof(b), "    %p  %s+0x%lx  (%s+0x%lx)\n", pc, info.dli_fname, (unsigned long)((char*)pc - (char*)info.dli_fbase),
                 info.dli_sname ? info.dli_sname : "?", info.dli_saddr ? (unsigned long)((char*)pc - (char*)info.dli_saddr) : 0ul);
    } else {
        snprintf(b, sizeof(b), "    %p  ?\n", pc);
    }
    put(b);
}

static void dump_maps(void) {
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) return;
    static char buf[1 << 16];
    static char line[1024];
    size_t fill = 0;
    ssize_t n;
    /* only the executable mappings of shared objects: enough to turn an address into library + offset */
    while ((n = read(fd, buf, sizeof(buf))) > 0) {
        for (ssize_t i = 0; i < n; i++) {
            if (fill < sizeof(line) - 1) line[fill++] = buf[i];
            if (buf[i] == '\n') {
                line[fill] = 0;
                if (strstr(line, " r-xp ") && strstr(line, ".so")) put(line);
                fill = 0;
            }
        }
    }
    close(fd);
}

static volatile int entered;

static void handler(int sig, siginfo_t* si, void* uc_) {
    ucontext_t* uc = (ucontext_t*)uc_;
    static char b[1024];
    if (entered++) { signal(sig, SIG_DFL); raise(sig); return; }
    snprintf(b, sizeof(b), "\n[segv_trace] signal %d, fault address %p, thread %ld\n", sig, si->si_addr, (long)syscall(SYS_gettid));
    put(b);
    greg_t* r = uc->uc_mcontext.gregs;
    snprintf(b, sizeof(b), "  rip %llx rsp %llx rax %llx rbx %llx rcx %llx rdx %llx rsi %llx rdi %llx\n  rbp %llx r8 %llx r9 %llx r10 %llx r11 %llx r12 %llx r13 %llx r14 %llx r15 %llx\n",
             (unsigned long long)r[REG_RIP], (unsigned long long)r[REG_RSP], (unsigned long long)r[REG_RAX], (unsigned long long)r[REG_RBX],
             (unsigned long long)r[REG_RCX], (unsigned long long)r[REG_RDX], (unsigned long long)r[REG_RSI], (unsigned long long)r[REG_RDI],
             (unsigned long long)r[REG_RBP], (unsigned long long)r[REG_R8], (unsigned long long)r[REG_R9], (unsigned long long)r[REG_R10],
             (unsigned long long)r[REG_R11], (unsigned long long)r[REG_R12], (unsigned long long)r[REG_R13], (unsigned long long)r[REG_R14],
             (unsigned long long)r[REG_R15]);
    put(b);
    /* what the stack holds (return addresses among it), readable without any unwinder */
    put("  stack words:");
    unsigned long long* sp = (unsigned long long*)r[REG_RSP];
    for (int i = 0; i < 96; i++) {
        snprintf(b, sizeof(b), "%s%llx", (i % 8) ? " " : "\n    ", sp[i]);
        put(b);
    }
    put("\n  executable mappings:\n");
    dump_maps();
    put("  call chain (addresses):\n");
    static void* frames[64];
    int n = backtrace(frames, 64);
    for (int i = 0; i < n; i++) { snprintf(b, sizeof(b), "    %p\n", frames[i]); put(b); }
    put("  call chain (symbols):\n");
    for (int i = 0; i < n; i++) where(frames[i]);
    signal(sig, SIG_DFL);
    raise(sig);
}

static char alt_stack[1 << 16];

/* (again) installs the handlers: callable from the host program once every other library has installed its own */
void segv_trace_install(void) {
    stack_t ss;
    ss.ss_sp = alt_stack; ss.ss_size = sizeof(alt_stack); ss.ss_flags = 0;
    sigaltstack(&ss, NULL);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
    sigaction(SIGABRT, &sa, NULL);
    void* warm[4];
    backtrace(warm, 4);      /* loads libgcc's unwinder now, not inside the handler */
}

__attribute__((constructor)) static void install(void) { segv_trace_install(); }
