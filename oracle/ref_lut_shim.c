/* oracle/ref_lut_shim.c -- TEST INFRASTRUCTURE.
 * Exposes the reference's own decode tables (src/dotp_lut.h:3,1033), compiled
 * from the header where it lies under /root/reference, so tests can pin the
 * oracle's bit decode against them.  Built only when /root/reference exists;
 * output goes to oracle/_ref/ (git-ignored). */
#include REF_DOTP_LUT_H
const double* ref_dotp_lut_a(void) { return dotp_lut_a; }
const double* ref_dotp_lut_b(void) { return dotp_lut_b; }
