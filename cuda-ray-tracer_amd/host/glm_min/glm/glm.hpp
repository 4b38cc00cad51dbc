// glm/glm.hpp -- minimal vector/matrix types for THIS repo's host code.
//
// The reference's back-end interface (include/update.h, include/scene.h, include/surface.h,
// include/light.h) is written in terms of glm types, and glm is neither vendored by the reference nor
// installed in this image.  This header provides just the subset that interface and our host adapter
// touch (SURVEY.md 8(b) "glm surface actually needed at the boundary"), with glm's names and glm's
// operation order, so that the same host sources also compile against a real glm when one is on the
// include path (put it before this directory).  It is product code for our own host side; it is never
// used to build anything from /root/reference.
#ifndef MI355RT_GLM_MIN_HPP
#define MI355RT_GLM_MIN_HPP

#include <cmath>
#include <cstddef>

namespace glm {

enum qualifier { packed_highp, defaultp = packed_highp };

template <int L, typename T, qualifier Q = defaultp>
struct vec;

template <typename T, qualifier Q>
struct vec<3, T, Q> {
    union { T x, r; };
    union { T y, g; };
    union { T z, b; };
    constexpr vec() : x(0), y(0), z(0) {}
    constexpr explicit vec(T s) : x(s), y(s), z(s) {}
    constexpr vec(T a, T b_, T c) : x(a), y(b_), z(c) {}
    template <typename U, qualifier P>
    constexpr vec(const vec<3, U, P> &v) : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z)) {}
    template <typename U, qualifier P>
    constexpr explicit vec(const vec<4, U, P> &v);
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec &operator+=(const vec &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec &operator-=(const vec &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec &operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
};

template <typename T, qualifier Q>
struct vec<4, T, Q> {
    union { T x, r; };
    union { T y, g; };
    union { T z, b; };
    union { T w, a; };
    constexpr vec() : x(0), y(0), z(0), w(0) {}
    constexpr explicit vec(T s) : x(s), y(s), z(s), w(s) {}
    constexpr vec(T a_, T b_, T c, T d) : x(a_), y(b_), z(c), w(d) {}
    constexpr vec(const vec<3, T, Q> &v, T d) : x(v.x), y(v.y), z(v.z), w(d) {}
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    const T &operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};

template <typename T, qualifier Q>
template <typename U, qualifier P>
constexpr vec<3, T, Q>::vec(const vec<4, U, P> &v)
    : x(static_cast<T>(v.x)), y(static_cast<T>(v.y)), z(static_cast<T>(v.z))
{}

#define GLM_MIN_V3 vec<3, T, Q>
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator+(const GLM_MIN_V3 &a, const GLM_MIN_V3 &b) { return GLM_MIN_V3(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator-(const GLM_MIN_V3 &a, const GLM_MIN_V3 &b) { return GLM_MIN_V3(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator*(const GLM_MIN_V3 &a, const GLM_MIN_V3 &b) { return GLM_MIN_V3(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator/(const GLM_MIN_V3 &a, const GLM_MIN_V3 &b) { return GLM_MIN_V3(a.x / b.x, a.y / b.y, a.z / b.z); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator*(const GLM_MIN_V3 &a, T s) { return GLM_MIN_V3(a.x * s, a.y * s, a.z * s); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator*(T s, const GLM_MIN_V3 &a) { return GLM_MIN_V3(s * a.x, s * a.y, s * a.z); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator/(const GLM_MIN_V3 &a, T s) { return GLM_MIN_V3(a.x / s, a.y / s, a.z / s); }
template <typename T, qualifier Q> constexpr GLM_MIN_V3 operator-(const GLM_MIN_V3 &a) { return GLM_MIN_V3(-a.x, -a.y, -a.z); }
template <typename T, qualifier Q> constexpr bool operator==(const GLM_MIN_V3 &a, const GLM_MIN_V3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
#undef GLM_MIN_V3

#define GLM_MIN_V4 vec<4, T, Q>
template <typename T, qualifier Q> constexpr GLM_MIN_V4 operator+(const GLM_MIN_V4 &a, const GLM_MIN_V4 &b) { return GLM_MIN_V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <typename T, qualifier Q> constexpr GLM_MIN_V4 operator-(const GLM_MIN_V4 &a, const GLM_MIN_V4 &b) { return GLM_MIN_V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
template <typename T, qualifier Q> constexpr GLM_MIN_V4 operator*(const GLM_MIN_V4 &a, const GLM_MIN_V4 &b) { return GLM_MIN_V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
template <typename T, qualifier Q> constexpr GLM_MIN_V4 operator*(const GLM_MIN_V4 &a, T s) { return GLM_MIN_V4(a.x * s, a.y * s, a.z * s, a.w * s); }
#undef GLM_MIN_V4

template <int C, int R, typename T, qualifier Q = defaultp>
struct mat;

// column-major 4x4, m[col][row]
template <typename T, qualifier Q>
struct mat<4, 4, T, Q> {
    typedef vec<4, T, Q> col_type;
    col_type value[4];
    constexpr mat() : value{col_type(1, 0, 0, 0), col_type(0, 1, 0, 0), col_type(0, 0, 1, 0), col_type(0, 0, 0, 1)} {}
    constexpr explicit mat(T s) : value{col_type(s, 0, 0, 0), col_type(0, s, 0, 0), col_type(0, 0, s, 0), col_type(0, 0, 0, s)} {}
    constexpr mat(const col_type &a, const col_type &b, const col_type &c, const col_type &d) : value{a, b, c, d} {}
    col_type &operator[](int i) { return value[i]; }
    const col_type &operator[](int i) const { return value[i]; }
};

// glm: (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
template <typename T, qualifier Q>
constexpr vec<4, T, Q> operator*(const mat<4, 4, T, Q> &m, const vec<4, T, Q> &v)
{
    return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w);
}

typedef vec<3, float> vec3;
typedef vec<3, double> dvec3;
typedef vec<3, int> ivec3;
typedef vec<4, float> vec4;
typedef vec<4, double> dvec4;
typedef mat<4, 4, double> dmat4;
typedef mat<4, 4, float> mat4;

template <typename T, qualifier Q> constexpr T dot(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <typename T, qualifier Q> constexpr T dot(const vec<4, T, Q> &a, const vec<4, T, Q> &b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
template <typename T, qualifier Q> constexpr T length2(const vec<3, T, Q> &a) { return dot(a, a); }
template <typename T, qualifier Q> inline vec<3, T, Q> normalize(const vec<3, T, Q> &v) { return v * (static_cast<T>(1) / std::sqrt(dot(v, v))); }
template <typename T, qualifier Q> constexpr vec<3, T, Q> cross(const vec<3, T, Q> &a, const vec<3, T, Q> &b)
{
    return vec<3, T, Q>(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
template <typename T> constexpr T min(T a, T b) { return (b < a) ? b : a; }
template <typename T> constexpr T max(T a, T b) { return (a < b) ? b : a; }
template <typename T, qualifier Q> constexpr vec<3, T, Q> min(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return vec<3, T, Q>(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
template <typename T, qualifier Q> constexpr vec<3, T, Q> max(const vec<3, T, Q> &a, const vec<3, T, Q> &b) { return vec<3, T, Q>(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
template <typename T> constexpr T radians(T deg) { return deg * static_cast<T>(0.01745329251994329576923690768489); }
using std::pow;

// glm::lookAt (right-handed), used by the host camera (src/ray-tracer.cpp:56)
template <typename T, qualifier Q>
inline mat<4, 4, T, Q> lookAt(const vec<3, T, Q> &eye, const vec<3, T, Q> &center, const vec<3, T, Q> &up)
{
    const vec<3, T, Q> f(normalize(center - eye));
    const vec<3, T, Q> s(normalize(cross(f, up)));
    const vec<3, T, Q> u(cross(s, f));
    mat<4, 4, T, Q> r(static_cast<T>(1));
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

// glm::inverse(mat4): cofactor expansion
template <typename T, qualifier Q>
inline mat<4, 4, T, Q> inverse(const mat<4, 4, T, Q> &m)
{
    T c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    T c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    T c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    T c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    T c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    T c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    typedef vec<4, T, Q> V;
    V f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11), f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    V v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]), v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    V i0(v1 * f0 - v2 * f1 + v3 * f2), i1(v0 * f0 - v2 * f3 + v3 * f4), i2(v0 * f1 - v1 * f3 + v3 * f5), i3(v0 * f2 - v1 * f4 + v2 * f5);
    V sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat<4, 4, T, Q> inv(i0 * sa, i1 * sb, i2 * sa, i3 * sb);
    V row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    V d0(m[0] * row0);
    T det = (d0.x + d0.y) + (d0.z + d0.w);
    T ood = static_cast<T>(1) / det;
    return mat<4, 4, T, Q>(inv[0] * ood, inv[1] * ood, inv[2] * ood, inv[3] * ood);
}

} // namespace glm

#endif
