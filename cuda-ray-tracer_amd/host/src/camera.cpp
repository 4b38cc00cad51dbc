// camera.cpp -- see camera.h.  Needs glm::lookAt and glm::inverse: provided by a real glm (<glm/gtc/matrix_transform.hpp>)
// when one is on the include path, otherwise by host/glm_min/glm/glm.hpp, which restates glm's published
// algorithms for both (right-handed lookAt, cofactor-expansion inverse).
#include "camera.h"

#include <cmath>

glm::dvec3 Camera::direction() const
{
    glm::dvec3 d;
    d.x = std::cos(glm::radians(yaw)) * std::cos(glm::radians(pitch));
    d.y = std::sin(glm::radians(pitch));
    d.z = std::sin(glm::radians(yaw)) * std::cos(glm::radians(pitch));
    return d;
}

glm::dmat4 Camera::matrix() const
{
    const glm::dvec3 up(0.0, 1.0, 0.0);
    const glm::dmat4 world_to_camera = glm::lookAt(position, position - direction(), up);
    return glm::inverse(world_to_camera);
}
