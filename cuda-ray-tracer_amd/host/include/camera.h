// camera.h -- the reference host's fly camera, headless.
//
// The reference keeps its camera as file-scope state of the interactive host (position, yaw, pitch;
// src/ray-tracer.cpp:25-37) and hands update() the matrix inverse(lookAt(position, position - direction, up))
// (src/ray-tracer.cpp:44-58).  This is the same computation as a value type, so that moving-camera sequences can
// be rendered, benchmarked and compared frame by frame without a window.  Start-up pose: position 0, yaw 90,
// pitch 0 -- which gives the identity matrix to ~6e-17.
#pragma once

#include <glm/glm.hpp>

struct Camera {
    glm::dvec3 position{0.0, 0.0, 0.0};
    double yaw = 90.0;   // degrees
    double pitch = 0.0;  // degrees

    // update_direction(), reference src/ray-tracer.cpp:44-52 (direction only; the strafing vectors are input handling)
    glm::dvec3 direction() const;
    // camera_matrix(), reference src/ray-tracer.cpp:54-58: camera-to-world, column-major
    glm::dmat4 matrix() const;
};
