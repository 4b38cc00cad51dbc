// update.h -- the three-function render back-end contract.
//
// API mirror of the reference's include/update.h:6-8.  A back end is chosen at link time by linking
// exactly one translation unit that defines these symbols (reference src/CMakeLists.txt:18-23).
// This repo's definition is ../src/update-hip.cpp, an adapter onto the C ABI of libmi355rt.so
// (include/mi355rt.h) -- it takes the place of src/update-cuda.cu.
#pragma once

#include "scene.h"

// Called once after the scene is loaded.  `texture` is the GL texture name the interactive host blits;
// the HIP back end renders into an offscreen device buffer and ignores it unless a presenter is
// installed (see update-hip.cpp).
void init_update(unsigned int texture, const Scene &scene);

// Renders one frame for the given camera-to-world matrix; returns the device render time in ms.
float update(const glm::dmat4 &camera_matrix);

// Releases device resources.
void cleanup_update();
