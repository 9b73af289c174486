// Which sources this library was built from: the Makefile passes the hash of csrc/*.hip, *.hpp (the value
// simspread_jl_amd._lib.source_hash() computes from the files) as SS_SOURCE_SHA.  The Python loader compares the two and
// refuses a library that is older than the sources next to it (a stale .so once produced a round's worth of wrong
// profiles).  This file is not part of the hash.
#include "simspread_hip.h"
#ifndef SS_SOURCE_SHA
#define SS_SOURCE_SHA "unknown"
#endif
extern "C" const char* ss_source_hash(void) { return SS_SOURCE_SHA; }
