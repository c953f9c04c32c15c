// 512-thread solve kernel of the default build in its own translation unit (namespace eicos::t512): see the note at the top of kernels.hip.
#define EICOS_TSPLIT 512
#include "kernels.hip"
