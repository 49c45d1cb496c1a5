// BMI2 + ADX build of the serial host-side group operations (mulx / adcx / adox carry
// chains halve the cost of the 64-bit-limb Montgomery product).  Compiled with
// -mbmi2 -madx; only called when the CPU reports both features.
#define curdle curdle_bmi2
#define CURDLE_ISA_SUFFIX _bmi2
#include "host_ops_impl.h"
