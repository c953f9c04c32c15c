// LDS-resident build of the solve kernel (namespace eicos::ldsres): see the note at the top of kernels.hip.
#define EICOS_LDSRES 1
#include "kernels.hip"
