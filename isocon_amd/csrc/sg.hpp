// sg.hpp -- semi-global affine alignment with traceback (kernels).  Filled in below.
#pragma once
#include "common.hpp"
