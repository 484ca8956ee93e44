// TEST INFRASTRUCTURE ONLY -- force-included (-include) when oracle/build_ref.sh compiles the
// reference's own sources (/root/reference/src/*.cpp) into oracle/_ref/.  It contains no
// reference code.  It provides exactly three things the reference needs or the tests need:
//
//  1. std::_Pi_val            -- an MSVC-STL-internal constant used by Render.cpp:73; libstdc++ has
//                                no such name.  Value = pi to double precision (what MSVC defines).
//  2. a seedable / injectable engine in place of `std::mt19937` inside utils.h:23-28 (rand1f).
//        mode 0: forwards to a genuine std::mt19937 (seeded 12345 instead of random_device, so
//                single-threaded runs are reproducible) -- stream semantics are unchanged.
//        mode 1: pops 32-bit words from a queue the test driver filled, so a KAT can dictate
//                every xi the reference consumes (cast_Ray, sample, BSDF::Sample, RR ...).
//  3. `mcpt_ref_max_bounces`  -- only read by the *depth* build variant, where build_ref.sh turns
//                                the unbounded `for (bounces = 0;; bounces++)` of Render.cpp:116
//                                into `bounces < mcpt_ref_max_bounces` in a temp copy.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <algorithm>
#include <limits>
#include <random>

namespace std { constexpr double _Pi_val = 3.14159265358979323846; }

namespace mcpt_refshim {
using genuine_mt19937 = std::mt19937;   // captured before the macro below renames the token
struct rng_control {
    int mode = 0;                 // 0 = genuine mt19937 stream, 1 = injected queue
    const uint32_t* queue = nullptr;
    long n = 0, pos = 0, underflow = 0;
};
extern rng_control g_rng;         // defined in ref_driver.cpp
extern int g_max_bounces;         // defined in ref_driver.cpp
}
static int& mcpt_ref_max_bounces = mcpt_refshim::g_max_bounces;

namespace std {
struct mcpt_fixed_random_device { unsigned operator()() { return 12345u; } };
struct mcpt_switchable_engine {
    typedef uint32_t result_type;
    mcpt_refshim::genuine_mt19937 g;
    explicit mcpt_switchable_engine(unsigned seed) : g(seed) {}
    static constexpr result_type min() { return 0u; }
    static constexpr result_type max() { return 0xffffffffu; }
    result_type operator()() {
        auto& c = mcpt_refshim::g_rng;
        if (c.mode == 0) return g();
        if (c.pos < c.n) return c.queue[c.pos++];
        c.underflow++;
        return 0u;
    }
};
}
#define random_device mcpt_fixed_random_device
#define mt19937 mcpt_switchable_engine
