// glm/gtx/norm.hpp -- length2() lives in our minimal glm.hpp (see there).
#ifndef MI355RT_GLM_MIN_NORM_HPP
#define MI355RT_GLM_MIN_NORM_HPP
#include "../glm.hpp"
#endif
