/*
 * ref_gr.c -- TEST INFRASTRUCTURE (oracle/): exposes the reference's `static` Gelman-Rubin
 * routine (check_converg.c:100-153) to oracle/ref_dump.c.  Built only in the development
 * container into oracle/_ref/ (the reference file is included by absolute path, not copied).
 */
#include "/root/reference/check_converg.c"
double ref_GelmanRubin(double *vec, int numchains, int totrep) { return GelmanRubin(vec, numchains, totrep); }
