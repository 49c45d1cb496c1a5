// BMI2 + ADX build of the window combine (mulx / adcx / adox carry chains halve
// the cost of the 64-bit-limb Montgomery product).  Compiled with -mbmi2 -madx;
// only called when the CPU reports both features.
#define curdle curdle_bmi2
#define CURDLE_COMBINE_NAME curdle_window_combine_bmi2
#include "window_combine_impl.h"
