// 128-thread solve kernel of the default build in its own translation unit (namespace eicos::t128): see the note at the top of kernels.hip.
#define EICOS_TSPLIT 128
#include "kernels.hip"
