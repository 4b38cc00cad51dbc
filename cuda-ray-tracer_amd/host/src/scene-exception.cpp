// scene-exception.cpp -- colour validator of the scene loader (reference src/scene-exception.cpp:3-11).
#include "scene-exception.h"

void validate_color(const glm::vec3 &color)
{
    const float ch[3] = {color.x, color.y, color.z};
    for (float v : ch) {
        if (v < 0.0f || v > 1.0f) {
            std::ostringstream msg;
            msg << "Invalid color: (" << color.x << ", " << color.y << ", " << color.z << ")";
            throw SceneException(msg.str());
        }
    }
}
