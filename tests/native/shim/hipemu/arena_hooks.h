// TEST INFRASTRUCTURE (tests/emu.py): under AddressSanitizer the padding behind every sub-allocation of a DevArena is
// poisoned, so that a kernel reading past one staged array into the next is caught.  Never part of libphant_gpu.so.
#pragma once
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define PHANT_ARENA_POISON(p, n) ASAN_POISON_MEMORY_REGION((p), (n))
#define PHANT_ARENA_UNPOISON(p, n) ASAN_UNPOISON_MEMORY_REGION((p), (n))
#define PHANT_ARENA_POISONS 1  // (a copy may then not span several sub-allocations: it would touch the padding)
#else
#define PHANT_ARENA_POISON(p, n) ((void)0)
#define PHANT_ARENA_UNPOISON(p, n) ((void)0)
#define PHANT_ARENA_POISONS 0
#endif
