// COMPILE-CHECK STUB (tools/stubs/README.md): the one macro the dump tool uses (reference: include/inria_wbc/exceptions.hpp:52-66).
#pragma once
#define IWBC_CHECK(expr) (expr)
