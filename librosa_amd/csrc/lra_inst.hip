// lra_inst.hip -- explicit instantiations of one group of fused kernels (see lra_fused.h).
// Compiled once per group: hipcc -c -DLRA_INST_GROUP=<k> lra_inst.hip -o inst_<k>.o
#include "lra_fused.h"

#ifndef LRA_INST_GROUP
#error "compile with -DLRA_INST_GROUP=<0..LRA_INST_NUM_GROUPS-1>"
#endif
#define LRA_CAT2(a, b) a##b
#define LRA_CAT(a, b) LRA_CAT2(a, b)
#if LRA_INST_GROUP == 12  // radix 16-16-4 second-generation kernels
LRA_INST2_GROUP_12(LRA_T_DEFINE, LRA_I_DEFINE)
#elif LRA_INST_GROUP == 11  // producer / consumer mel kernels
LRA_CAT(LRA_INST3_GROUP_, LRA_INST_GROUP)(LRA_P_DEFINE)
#elif LRA_INST_GROUP >= 9  // second-generation forward kernels
LRA_CAT(LRA_INST2_GROUP_, LRA_INST_GROUP)(LRA_T_DEFINE, LRA_I_DEFINE)
#else
LRA_CAT(LRA_INST_GROUP_, LRA_INST_GROUP)(LRA_S_DEFINE, LRA_I_DEFINE)
#endif
