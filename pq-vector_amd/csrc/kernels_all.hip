// kernels_all.hip -- every kernel translation unit in ONE (diagnostic builds only: `make phases`, `make stamps` -- the stamps
// build keeps its device-side stamp table in one place).  The product build compiles the seven units separately.
#include "kernels_probe.hip"
#include "kernels_screen.hip"
#include "kernels_brute.hip"
#include "kernels_build.hip"
#include "kernels_layout.hip"
#include "kernels_list.hip"
#include "kernels_kpp.hip"
