/* Stand-in for the header of the same name that the reference's tests/unit.c includes (line 4) for its
 * closing printout (lines 2824-2834).  That header is the reference's CPU feature probe; this library
 * has none to offer (it dispatches on the pointer kind and the call's size, not on CPUID), so the
 * stand-in defines neither STREAMVBYTE_X64 nor streamvbyte_sse41() and unit.c's printout takes its
 * "non-vectorized" branch.  Used only by tests/test_reference_programs.py. */
#ifndef STREAMVBYTE_ISADETECTION_STANDIN_H
#define STREAMVBYTE_ISADETECTION_STANDIN_H
#endif
