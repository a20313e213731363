// Registry of the kernel instantiations the library can select (sesrq_common.h: KernelInstance, launch_kernel) and its C ABI
// (include/sesrq.h: sesrq_instance_count / _name / _launches).  Filled by the dynamic initialisers of KernelInstance<KERN>::id when the
// library is loaded; read-only afterwards except for the launch counters.
#include <cxxabi.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <deque>
#include <mutex>
#include <string>

#include "sesrq_common.h"

namespace sesrq {
namespace {
struct Entry {
    std::string name;
    const void *fn;
    std::atomic<long long> launches{0};
    Entry(std::string n, const void *f) : name(std::move(n)), fn(f) {}
};
struct Registry {
    std::mutex mu;
    std::deque<Entry> entries;      // deque: growth never moves an element (the atomics are addressed while others register)
};
Registry &registry() { static Registry r; return r; }      // constructed on first use: initialisation order of the translation units does not matter
}  // namespace

// The name of an instantiation: __PRETTY_FUNCTION__ prints a pointer-to-function template argument without ITS template arguments
// ("&sesrq::mfma_h5_kernel"), so the name comes from the host stub's own symbol -- the kernels are namespace-scope templates, their stubs
// weak symbols of the library's dynamic symbol table: dladdr + the C++ demangler give "void sesrq::__device_stub__mfma_h5_kernel<1, 2, 22,
// 3>(sesrq::ConvArgs)", the same spelling a rocprofv3 kernel trace prints.  Fallback: the pretty-function text + the stub's address.
int register_instance(const void *host_fn, const char *pretty_function) {
    std::string s;
    Dl_info info;
    if (dladdr(host_fn, &info) && info.dli_sname && info.dli_saddr == host_fn) {
        int status = 0;
        char *dm = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
        s = (status == 0 && dm) ? dm : info.dli_sname;
        free(dm);
        if (s.compare(0, 5, "void ") == 0) s.erase(0, 5);
        // cut the parameter list: the last '(' at template depth 0
        int depth = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == '<') ++depth;
            else if (s[i] == '>') --depth;
            else if (s[i] == '(' && depth == 0) { s.erase(i); break; }
        }
    } else {
        s = pretty_function;
        const size_t a = s.rfind("KERN = ");
        if (a != std::string::npos) {
            s = s.substr(a + 7);
            if (!s.empty() && s.back() == ']') s.pop_back();
            if (!s.empty() && s[0] == '&') s.erase(0, 1);
        }
        char buf[32];
        snprintf(buf, sizeof(buf), "@%p", host_fn);
        s += buf;
    }
    if (getenv("SESRQ_REG_DEBUG")) fprintf(stderr, "reg %p %s | %s\n", host_fn, s.c_str(), pretty_function);
    for (size_t p; (p = s.find("sesrq::")) != std::string::npos;) s.erase(p, 7);
    for (size_t p; (p = s.find("__device_stub__")) != std::string::npos;) s.erase(p, 15);
    for (size_t p; (p = s.find("(anonymous namespace)::")) != std::string::npos;) s.erase(p, 23);
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    for (size_t i = 0; i < r.entries.size(); ++i)
        if (r.entries[i].fn == host_fn) return (int)i;
    r.entries.emplace_back(s, host_fn);
    return (int)r.entries.size() - 1;
}

void count_launch(int id) {
    Registry &r = registry();
    if (id >= 0 && (size_t)id < r.entries.size()) r.entries[(size_t)id].launches.fetch_add(1, std::memory_order_relaxed);
}

}  // namespace sesrq

using namespace sesrq;

extern "C" {

int sesrq_instance_count(void) {
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    return (int)r.entries.size();
}

const char *sesrq_instance_name(int i) {
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    return (i >= 0 && (size_t)i < r.entries.size()) ? r.entries[(size_t)i].name.c_str() : "";
}

long long sesrq_instance_launches(int i) {
    Registry &r = registry();
    std::lock_guard<std::mutex> lk(r.mu);
    return (i >= 0 && (size_t)i < r.entries.size()) ? r.entries[(size_t)i].launches.load(std::memory_order_relaxed) : -1;
}

}  // extern "C"
