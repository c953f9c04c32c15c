// kernels.hip with the factor operand array U resident in LDS, 256 threads (namespace eicos::ubl256): see EICOS_UBL in kernels.hip
#define EICOS_UBL 256
#include "kernels.hip"
