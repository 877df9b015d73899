/* TEST INFRASTRUCTURE: see Rinternals.h in this directory */
#ifndef POTUS_R_STUB_R_H
#define POTUS_R_STUB_R_H
#endif
