// COMPILE-CHECK STUB (tools/stubs/README.md).  Not tsid.
#pragma once
