// 256-thread solve kernel compiled for two waves per SIMD (namespace eicos::w2): see the note at the top of kernels.hip.
#define EICOS_W2 1
#include "kernels.hip"
