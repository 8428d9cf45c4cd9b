// A region allocator for what one PlanNextMapEx call hands to its caller (blance_api.cpp): the million Partition objects
// of the result, their maps, lists and map nodes -- ~7 M small allocations at BASELINE config 3, which is where the C++
// mirror's time went (glibc: ~25 ns per malloc, as much again per free, and a coalescing pass over the freed chunks).
// The Go shim gets the same effect from `make([]Partition, P)` + one backing slice (go/blance/plan_hip.go).
//
// The mirror keeps the reference's types (std::map for Go's map, std::string, std::vector), so the region sits behind
// the global operator new / delete (call_arena.cpp; linked into a program only if it wants it):
//   * inside an arena::Scope, on the thread that opened it, `new` is a pointer bump in the thread's current 4 MB chunk;
//     everywhere else it is malloc, as before;
//   * `delete` of a pointer inside the reserved address range drops the live count of the pointer's chunk (one atomic
//     decrement); a chunk whose count reaches zero goes back to the pool and is reused, pages warm, by a later call.
//     So a caller may drop the result piecemeal, from any thread, at any time -- no lifetime rule is added to the API;
//   * anything larger than a quarter of a chunk, or over-aligned, is malloc'ed as before.
// arena::trim() returns the pooled chunks' pages to the system (Library::trim calls it).
#pragma once
#include <stddef.h>

namespace blance {
namespace arena {

bool available();                 // the address range could be reserved and call_arena.cpp is linked in

struct Scope {                    // allocations of THIS thread between construction and destruction come from the region
    Scope();
    ~Scope();
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;
  private:
    bool outer_;
};

struct Stats {
    size_t chunks_in_use, chunks_pooled, chunks_ever;
};
Stats stats();
void trim();

}  // namespace arena
}  // namespace blance
