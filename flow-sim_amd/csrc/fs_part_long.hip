// kernel instantiations: reaches longer than one lane grid (fs_long.hpp)
#include "fs_entries.hpp"
FS_LIST_LONG(FS_INSTANTIATE_LONG)
