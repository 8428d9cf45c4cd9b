// Exercises blance_amd/csrc/host/call_arena.cpp on its own: objects made inside arena::Scope on several threads, deleted
// piecemeal from other threads, chunks recycled and reused, oversized and over-aligned requests passed through to malloc,
// nothing from the region outside a Scope.  Prints "ok" or a line saying what went wrong.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../blance_amd/csrc/host/call_arena.hpp"

using namespace blance;

struct alignas(64) Wide { char c[64]; };

static int fail(const char* what) { printf("FAIL %s\n", what); return 1; }

int main() {
    if (!arena::available()) { printf("ok (no address range: arena off)\n"); return 0; }
    const arena::Stats s0 = arena::stats();
    if (s0.chunks_in_use != 0) return fail("chunks in use before anything ran");
    { std::unique_ptr<int> outside(new int(7)); if (arena::stats().chunks_in_use != 0) return fail("allocation outside a Scope took a chunk"); }
    // build on 4 threads, hand over, delete on 4 other threads in another order
    std::vector<std::vector<std::map<std::string, std::vector<std::string>>*>> made(4);
    for (auto& v : made) v.reserve(20000);                   // (outside any Scope: the lists themselves are malloc'ed)
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++)
            th.emplace_back([&, t]() {
                arena::Scope scope;
                for (int i = 0; i < 20000; i++) {
                    auto* m = new std::map<std::string, std::vector<std::string>>();
                    (*m)["primary-state-name-longer-than-sso"] = {"n" + std::to_string(i), std::string(40, 'x')};
                    (*m)["replica"] = {"a", "b"};
                    made[(size_t)t].push_back(m);
                }
                std::unique_ptr<char[]> big(new char[3u << 20]);      // larger than a quarter chunk: malloc'ed
                memset(big.get(), 1, 3u << 20);
                std::unique_ptr<Wide> wide(new Wide());               // over-aligned: malloc'ed
                if (((uintptr_t)wide.get() & 63) != 0) { printf("FAIL alignment\n"); exit(1); }
            });
        for (auto& x : th) x.join();
    }
    const arena::Stats s1 = arena::stats();
    if (s1.chunks_in_use < 4) return fail("the threads' objects are not in chunks");
    for (auto& v : made) for (auto* m : v) if ((*m)["replica"].size() != 2 || (*m)["primary-state-name-longer-than-sso"][1].size() != 40) return fail("content");
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++)
            th.emplace_back([&, t]() {
                auto& v = made[(size_t)((t + 1) % 4)];
                for (size_t i = v.size(); i-- > 0;) delete v[i];        // reverse order, another thread
            });
        for (auto& x : th) x.join();
    }
    const arena::Stats s2 = arena::stats();
    if (s2.chunks_in_use != 0) { printf("FAIL %zu chunks still in use after everything was deleted\n", s2.chunks_in_use); return 1; }
    if (s2.chunks_pooled != s1.chunks_ever) return fail("chunks not pooled");
    // a second round reuses the pooled chunks instead of fresh ones
    {
        arena::Scope scope;
        std::vector<std::unique_ptr<std::string>> v;
        for (int i = 0; i < 100000; i++) v.emplace_back(new std::string(100, 'y'));
        if (arena::stats().chunks_ever != s1.chunks_ever) return fail("fresh chunks although the pool had some");
        {
            arena::Scope inner;                                         // scopes nest
            v.emplace_back(new std::string(200, 'z'));
        }
        v.emplace_back(new std::string(300, 'w'));
        if (arena::stats().chunks_in_use == 0) return fail("nested scope closed the outer one");
    }
    arena::trim();
    {
        arena::Scope scope;                                             // after trim the pooled chunks still work
        std::unique_ptr<std::string> s(new std::string(1000, 'q'));
        if ((*s)[999] != 'q') return fail("after trim");
    }
    printf("ok\n");
    return 0;
}
