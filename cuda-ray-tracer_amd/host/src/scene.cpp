// scene.cpp -- Scene / Object constructors and the YAML scene loader.
//
// Behavioural mirror of the reference loader (src/scene.cpp:6-203; format description
// presentation/Instrukcja.md:17-33): same keys, same defaults (max_reflections 5, bg_color WHITE,
// reflection_ratio 0, sphere centre 0 / radius 1, plane origin 0 / normal +y, light intensity 1 / colour
// white), same mandatory keys, same error texts including the "line: N column: M" suffix
// (src/scene.cpp:24-39).  The YAML syntax layer is this repo's own (yaml_subset.*).
#include "scene.h"

#include <string>

#include "scene-exception.h"
#include "yaml_subset.h"

using yamlsub::Node;

namespace {

// loader defaults, reference src/scene.cpp:6-7
const unsigned int kDefaultMaxReflections = 5;
const glm::vec3 kDefaultBackground(1.0f);

std::string where(const yamlsub::Mark &m)
{
    return "line: " + std::to_string(m.line + 1) + " column: " + std::to_string(m.column + 1);
}

SceneException undefined_value(const Node &parent, const char *key)
{
    return SceneException(std::string("Value '") + key + "' undefined, " + where(parent.mark()));
}

SceneException invalid_value(const Node &value, const char *key)
{
    return SceneException(std::string("Value '") + key + "' is invalid, " + where(value.mark()));
}

// conversions of a node to the C++ type asked for; vectors are 3-element sequences of scalars
bool convert(const Node &n, double &out) { return n.to(out); }
bool convert(const Node &n, float &out) { return n.to(out); }
bool convert(const Node &n, unsigned int &out) { return n.to(out); }
bool convert(const Node &n, std::string &out) { return n.to(out); }

// a vector element that is not a number: yaml-cpp's inner as<T>() throws BadConversion carrying the ELEMENT's mark
// (reference src/scene.cpp:84-92)
struct BadElement {
    const Node *element;
};

template <typename T, glm::qualifier Q>
bool convert(const Node &n, glm::vec<3, T, Q> &out)
{
    if (!n.is_sequence() || n.size() != 3) return false;
    T v[3];
    for (int i = 0; i < 3; i++)
        if (!n[(size_t) i].to(v[i])) throw BadElement{&n[(size_t) i]};
    out = glm::vec<3, T, Q>(v[0], v[1], v[2]);
    return true;
}

// mandatory key: "undefined" if absent, "is invalid" if it does not convert.  For a vector with a bad element the reference
// reports the same text with the element's mark: get_value catches the BadConversion of the inner as<T>() (src/scene.cpp:48-53).
template <typename T>
T required(const Node &parent, const char *key)
{
    const Node &n = parent[key];
    if (!n.defined()) throw undefined_value(parent, key);
    T out{};
    try {
        if (!convert(n, out)) throw invalid_value(n, key);
    } catch (const BadElement &b) {
        throw invalid_value(*b.element, key);
    }
    return out;
}

// optional key: the default when absent AND when present but not convertible (yaml-cpp's as<T>(fallback)).  One case differs
// by design: a vector whose element is not a number.  In the reference the inner as<T>() throws a yaml-cpp BadConversion that
// as<T>(fallback) does not catch and main() does not either (it catches SceneException only, src/ray-tracer.cpp:151-158), so
// the program terminates; here it is reported as a SceneException with the element's position.
template <typename T>
T optional(const Node &parent, const char *key, const T &fallback)
{
    const Node &n = parent[key];
    if (!n.defined()) return fallback;
    T out{};
    try {
        return convert(n, out) ? out : fallback;
    } catch (const BadElement &b) {
        throw SceneException(std::string("Vector component of '") + key + "' is invalid, " + where(b.element->mark()));
    }
}

const Node &required_sequence(const Node &parent, const char *key)
{
    const Node &n = parent[key];
    if (!n.defined()) throw undefined_value(parent, key);
    if (!n.is_sequence()) throw SceneException(std::string("Value '") + key + "' must be a sequence, " + where(n.mark()));
    return n;
}

const Node &required_mapping(const Node &parent, const char *key)
{
    const Node &n = parent[key];
    if (!n.defined()) throw undefined_value(parent, key);
    if (!n.is_map()) throw SceneException(std::string("Value '") + key + "' must be a mapping, " + where(n.mark()));
    return n;
}

// object "type" -> coefficients, reference src/scene.cpp:97-151
SurfaceCoefs surface_from(const Node &node)
{
    const std::string type = required<std::string>(node, "type");
    if (type == "sphere")
        return SurfaceCoefs::sphere(optional(node, "center", glm::dvec3(0.0)), optional(node, "radius", 1.0));
    if (type == "plane")
        return SurfaceCoefs::plane(optional(node, "origin", glm::dvec3(0.0)), optional(node, "normal", glm::dvec3(0.0, 1.0, 0.0)));
    if (type == "dingDong") return SurfaceCoefs::dingDong(optional(node, "origin", glm::dvec3(0.0)));
    if (type == "clebsch") return SurfaceCoefs::clebsch();
    if (type == "cayley") return SurfaceCoefs::cayley();
    if (type == "polynomial") {
        const Node &table = required_mapping(node, "coefficients");
        static const char *names[20] = {"x3", "y3", "z3", "x2y", "xy2", "x2z", "xz2", "y2z", "yz2", "xyz",
                                        "x2", "y2", "z2", "xy", "xz", "yz", "x", "y", "z", "c"};
        SurfaceCoefs s{};
        double *dst = s.data();
        for (int i = 0; i < 20; i++) dst[i] = optional(table, names[i], 0.0); // omitted coefficients are 0
        return s;
    }
    throw SceneException("Unknown surface type: '" + type + "', " + where(node["type"].mark()));
}

LightSource light_from(const Node &node)
{
    const std::string type = required<std::string>(node, "type");
    const float intensity = optional(node, "intensity", 1.0f);
    if (type == "directional")
        return LightSource::directional(intensity, required<glm::dvec3>(node, "direction"), optional(node, "color", glm::vec3(1.0f)));
    if (type == "spherical")
        return LightSource::spherical(intensity, required<glm::dvec3>(node, "position"), optional(node, "color", glm::vec3(1.0f)));
    throw SceneException("Light source type must be 'spherical' or 'directional', " + where(node["type"].mark()));
}

} // namespace

Object::Object(SurfaceCoefs surface_, float reflection_ratio_, const glm::vec3 &color_)
    : surface(surface_), reflection_ratio(reflection_ratio_), color(color_)
{
    validate_positive("object reflection ratio", reflection_ratio);
    validate_color(color);
}

Scene::Scene(unsigned int width, unsigned int height, double vertical_fov_deg, unsigned int max_reflections_,
             const glm::vec3 &bg)
    : px_width(width), px_height(height), vertical_fov(glm::radians(vertical_fov_deg)), bg_color(bg),
      max_reflections(max_reflections_)
{
    validate_color(bg_color);
}

Scene Scene::load_from_file(const char *path)
{
    Node root;
    try {
        root = yamlsub::load_file(path);
    } catch (const yamlsub::FileError &) {
        throw SceneException(std::string("Cannot read the file ") + path);
    } catch (const yamlsub::ParseError &e) {
        throw SceneException(std::string("YAML parser error: ") + e.what());
    }

    // evaluated one by one so that the first missing key reported is deterministic
    const unsigned int width = required<unsigned int>(root, "width");
    const unsigned int height = required<unsigned int>(root, "height");
    const double fov_deg = required<double>(root, "fov");
    Scene scene(width, height, fov_deg, optional(root, "max_reflections", kDefaultMaxReflections),
                optional(root, "bg_color", kDefaultBackground));
    const Node &objects = required_sequence(root, "objects");
    const Node &lights = required_sequence(root, "light_sources");
    for (const Node &node : objects.items()) {
        SurfaceCoefs surface = surface_from(node);
        float reflection = optional(node, "reflection_ratio", 0.0f);
        scene.objects.push_back(Object(surface, reflection, required<glm::vec3>(node, "color")));
    }
    for (const Node &node : lights.items()) scene.lights.push_back(light_from(node));
    return scene;
}
