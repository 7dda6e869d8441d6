// mdx_nb_inst.hip - one translation unit per (ENERGY, COUL) pair of the pair kernels: the Makefile compiles this file eight times
// with -DNB_TU_ENERGY=0|1 -DNB_TU_COUL=0..3 (CM_SHIFTED, CM_RF, CM_EWALD, CM_SOFT) and once with -DNB_TU_ENERGY=0 -DNB_TU_COUL=4
// (CM_EWALD_TAB: the force-only Ewald flavour with the table).
#include "mdx_nonbonded_impl.h"
#if !defined(NB_TU_ENERGY) || !defined(NB_TU_COUL)
#error "compile with -DNB_TU_ENERGY=0|1 -DNB_TU_COUL=0..3"
#endif
template void launch_variant<(NB_TU_ENERGY != 0), NB_TU_COUL>(mdx_handle*, const NbArgs&, bool, bool);
