// o3dr_profile.h — optional per-kernel HIP-event bracketing (bench.py's roofline leg).
// Disabled by default: no event is recorded unless o3dr_profile_enable() asked for that kernel id.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/o3dr.h"

namespace o3dr {

struct Profiler {
    uint32_t mask = 0;  // bit k: bracket launches of kernel id k
    struct Pair {
        hipEvent_t a, b;
    };
    std::vector<Pair> open[O3DR_K_NUM];  // recorded, not yet read
    std::vector<Pair> pool;              // recycled
    double total_ms[O3DR_K_NUM] = {0};
    int64_t launches[O3DR_K_NUM] = {0};

    Pair acquire()
    {
        if (!pool.empty()) {
            Pair p = pool.back();
            pool.pop_back();
            return p;
        }
        Pair p;
        (void)hipEventCreate(&p.a);
        (void)hipEventCreate(&p.b);
        return p;
    }
    // fold finished pairs into the totals (caller has synchronised the stream)
    void drain()
    {
        for (int k = 0; k < O3DR_K_NUM; ++k) {
            for (Pair& p : open[k]) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                    total_ms[k] += ms;
                    launches[k] += 1;
                }
                pool.push_back(p);
            }
            open[k].clear();
        }
    }
    void reset()
    {
        drain();
        for (int k = 0; k < O3DR_K_NUM; ++k) total_ms[k] = 0, launches[k] = 0;
    }
    ~Profiler()
    {
        for (int k = 0; k < O3DR_K_NUM; ++k)
            for (Pair& p : open[k]) pool.push_back(p);
        for (Pair& p : pool) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    }
};

struct ProfScope {
    Profiler* pf;
    int kid;
    hipStream_t s;
    Profiler::Pair p;
    bool on;
    ProfScope(Profiler* pf_, int kid_, hipStream_t s_) : pf(pf_), kid(kid_), s(s_), on(false)
    {
        if (pf && (pf->mask >> kid) & 1u) {
            p = pf->acquire();
            (void)hipEventRecord(p.a, s);
            on = true;
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(p.b, s);
            pf->open[kid].push_back(p);
        }
    }
};

}  // namespace o3dr
