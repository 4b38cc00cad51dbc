// scene-exception.h -- error type and validators of the scene loader.
//
// API mirror of the reference's include/scene-exception.h:10-36 (SceneException, validate_positive,
// validate_color: same names, same message texts).
#pragma once

#include <exception>
#include <sstream>
#include <string>
#include <utility>

#include <glm/glm.hpp>

class SceneException : public std::exception {
public:
    explicit SceneException(std::string message) : message_(std::move(message)) {}
    const char *what() const noexcept override { return message_.c_str(); }

private:
    std::string message_;
};

// Rejects values below zero only (zero passes), reference include/scene-exception.h:26-34.
template <typename T>
void validate_positive(const char *what, const T &value)
{
    if (value < 0) {
        std::ostringstream msg;
        msg << "Negative value for " << what << ": " << value;
        throw SceneException(msg.str());
    }
}

// Every channel must lie in [0, 1], reference src/scene-exception.cpp:3-11.
void validate_color(const glm::vec3 &color);
