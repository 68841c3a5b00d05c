// kernel instantiations of libflowsim_hip.so, part "table_f64" (see fs_entries.hpp)
#include "fs_entries.hpp"

FS_LIST_TABLE(FS_INSTANTIATE, double, FS_F64)
