// canopy_guard.h -- no C++ exception crosses the C ABI: every extern "C" entry point that can allocate
// (std::string / std::vector / new) is a function-try-block closed by CANOPY_CATCH_ALL.
#pragma once

#include <new>

#include "canopy.h"

#define CANOPY_CATCH_ALL                                              \
    catch (const std::bad_alloc&) { return CANOPY_ERR_OUT_OF_MEMORY; } \
    catch (...) { return CANOPY_ERR_INTERNAL; }
