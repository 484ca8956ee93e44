// CPU-only check of the Scene <-> FilmSource protocol of host/Scene.{h,cpp} (built with -fsanitize=address by tests/test_abi_and_host.py):
// two sources take turns on one Scene, and the Scene dies before either of them.  `Holder` keeps the bookkeeping host/Render.cpp keeps
// (a `target` it flushes into and detaches from in its destructor); the displaced one must have been told to forget the Scene, or its
// destructor calls detach() on freed memory.
#include <cstdio>
#include <memory>
#include "Scene.h"

struct Holder : FilmSource {
    Scene* target = nullptr; int pending = 0, flushed = 0, gone = 0, displaced_n = 0;
    void add(Scene& s) { if (target != &s) { if (target) { flush_into(*target); target->detach(this); } target = &s; } s.attach(this); pending++; }
    void flush_into(Scene& s) override { if (&s != target) return; flushed += pending; pending = 0; }
    void scene_gone(Scene& s) override { if (&s == target) { target = nullptr; gone++; pending = 0; } }
    void displaced(Scene& s) override { if (&s == target) { target = nullptr; displaced_n++; } }
    ~Holder() override { if (target) { flush_into(*target); target->detach(this); } }
};

int main() {
    auto a = std::make_unique<Holder>(); auto b = std::make_unique<Holder>();
    {
        auto s = std::make_unique<Scene>(4, 4);
        a->add(*s); a->add(*s);
        b->add(*s);                                       // takes the Scene over: a is flushed, then displaced
        if (a->flushed != 2 || a->displaced_n != 1 || a->target != nullptr || s->source() != b.get()) return 1;
        a->add(*s);                                       // and back: b flushed + displaced
        if (b->flushed != 1 || b->displaced_n != 1 || b->target != nullptr || s->source() != a.get()) return 2;
        b->add(*s);
        s.reset();                                        // the Scene dies first: only b (the attached one) hears scene_gone
        if (b->gone != 1 || b->target != nullptr || a->target != nullptr) return 3;
    }
    a.reset(); b.reset();                                 // neither destructor may touch the dead Scene
    std::puts("ok");
    return 0;
}
