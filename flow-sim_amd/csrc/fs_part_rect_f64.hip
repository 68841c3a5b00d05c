// kernel instantiations of libflowsim_hip.so, part "rect_f64" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_RECT(FS_INSTANTIATE, double, FS_F64)
