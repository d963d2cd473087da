// No C++ exception crosses the C ABI ("no exceptions across the boundary").
//
// The reference's path is infallible (in_memory.rs:72-156 always answers Ok) and its errors are VALUES
// (StorageErr, storage/mod.rs:312-339) — a host that links these libraries must never be taken down by a
// std::bad_alloc / std::length_error / std::system_error thrown behind an entry point (the host side allocates
// std::vector / std::string / unordered_map storage behind most of them).  Every `extern "C"` entry whose body can
// allocate is a function-try-block that ends in RL_ABI_CATCH: the exception is turned into RL_ERR_NOMEM
// (std::bad_alloc) or RL_ERR_INTERNAL (anything else), what() is kept per thread for rl_last_internal_error(), and
// RAII (the engine's lock_guard, vectors) unwinds normally.  The engine's table is only ever written by kernels a call
// has already validated, so an exception on the host side of a call leaves the counters as they were or as the batch
// made them — never half a host-side structure.
#ifndef RL_ABI_GUARD_H
#define RL_ABI_GUARD_H

#include <exception>
#include <new>

#include "rl_engine.h"

// (exported by librl_engine.so; librl_storage.so / librl_sharded.so link against it)
extern "C" int32_t rl_abi_caught(const char* fn, const char* what, int32_t status);

#define RL_ABI_CATCH                                                                              \
    catch (const std::bad_alloc&) {                                                               \
        return rl_abi_caught(__func__, "std::bad_alloc (host memory exhausted)", RL_ERR_NOMEM);   \
    }                                                                                             \
    catch (const std::exception& rl_abi_ex) {                                                     \
        return rl_abi_caught(__func__, rl_abi_ex.what(), RL_ERR_INTERNAL);                        \
    }                                                                                             \
    catch (...) {                                                                                 \
        return rl_abi_caught(__func__, "unknown C++ exception", RL_ERR_INTERNAL);                 \
    }

#endif
