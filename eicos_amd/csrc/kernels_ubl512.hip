// kernels.hip with the factor operand array U resident in LDS, 512 threads (namespace eicos::ubl512): see EICOS_UBL in kernels.hip
#define EICOS_UBL 512
#include "kernels.hip"
