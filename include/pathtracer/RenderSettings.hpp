// Same name as the reference header; the whole API lives in pathtracer.hpp.
#pragma once
#include "pathtracer.hpp"
