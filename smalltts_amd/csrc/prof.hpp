// Optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
// Disabled by default: when off, prof_begin/prof_end are a null-pointer test.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

extern thread_local const char* g_prof_tag;

struct ProfAgg {
    long launches = 0;
    double ms = 0, flops = 0, bytes = 0;
    double bytes8d = 0;  // SURVEY 8(d) accounting: GEMMs = their weights once at 2 B / parameter (activations "negligible"), other kernels = bytes
};

class Profiler {
  public:
    ~Profiler() { reset(); }
    void begin(hipStream_t st, const std::string& name, double flops, double bytes, double bytes8d) {
        Rec r;
        r.name = (g_prof_tag && *g_prof_tag) ? std::string(g_prof_tag) + "/" + name : name;
        r.flops = flops; r.bytes = bytes; r.bytes8d = bytes8d < 0 ? bytes : bytes8d;
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, st);
        recs_.push_back(r);
    }
    void end(hipStream_t st) {
        if (!recs_.empty()) (void)hipEventRecord(recs_.back().b, st);
    }
    // call after the stream is synchronised
    std::map<std::string, ProfAgg> collect() {
        std::map<std::string, ProfAgg> out;
        for (auto& r : recs_) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                ProfAgg& a = out[r.name];
                a.launches++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.bytes8d += r.bytes8d;
            }
        }
        return out;
    }
    void reset() {
        for (auto& r : recs_) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        recs_.clear();
    }

  private:
    struct Rec { std::string name; double flops, bytes, bytes8d; hipEvent_t a, b; };
    std::vector<Rec> recs_;
};

extern thread_local Profiler* g_prof;  // set by Engine while profiling is enabled
extern thread_local const char* g_prof_tag;  // non-null in tagged mode: pipeline phase prefixed to every kernel name
struct ProfTag {  // scoped phase label ("enc", "dit", "dec.s3" ...); no effect unless tagged mode is on
    const char* prev;
    explicit ProfTag(const char* t) : prev(g_prof_tag) { if (g_prof_tag) g_prof_tag = t; }
    ~ProfTag() { if (g_prof_tag) g_prof_tag = prev; }
};

struct ProfScope {
    hipStream_t st;
    bool on;
    ProfScope(hipStream_t s, const char* name, double flops, double bytes, double bytes8d = -1.0) : st(s), on(g_prof != nullptr) {
        if (on) g_prof->begin(st, name, flops, bytes, bytes8d);
    }
    ProfScope(hipStream_t s, const std::string& name, double flops, double bytes, double bytes8d = -1.0) : st(s), on(g_prof != nullptr) {
        if (on) g_prof->begin(st, name, flops, bytes, bytes8d);
    }
    ~ProfScope() { if (on) g_prof->end(st); }
};
