// yaml_subset.h -- the part of YAML the scene format needs, parsed into a small DOM.
//
// The reference reads scenes through yaml-cpp (src/scene.cpp:3,154-203), which this repo does not
// depend on.  Supported: comments, block mappings, block sequences (also "- key: value" items and
// sequences indented like their parent key), flow sequences / flow mappings (nesting, inner spaces,
// spanning lines), plain / single- / double-quoted scalars, an optional leading "---", files without a
// trailing newline.  Not supported (rejected with a parse error): anchors/aliases, tags, block scalars
// (| and >), multi-document streams, tab indentation.
//
// Every node remembers where it starts (0-based line/column, like YAML::Mark) so the loader can word
// its errors the way the reference does (src/scene.cpp:24-39).  Scalars stay text until converted;
// the conversions follow yaml-cpp's rules (whole token must convert, .inf/.nan spellings, no negative
// text for unsigned).
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace yamlsub {

struct Mark {
    int line = 0, column = 0;
};

struct ParseError : std::runtime_error {
    Mark mark;
    ParseError(const Mark &m, const std::string &msg);
};

class Node {
public:
    enum Type { Undefined, Null, Scalar, Sequence, Map };

    Node() = default;
    Type type() const { return type_; }
    bool defined() const { return type_ != Undefined; }
    bool is_scalar() const { return type_ == Scalar; }
    bool is_sequence() const { return type_ == Sequence; }
    bool is_map() const { return type_ == Map; }
    const Mark &mark() const { return mark_; }
    const std::string &scalar() const { return text_; }

    size_t size() const { return type_ == Sequence ? items_.size() : (type_ == Map ? keys_.size() : 0); }
    // sequence element / map value; an Undefined node when absent (like YAML::Node::operator[])
    const Node &operator[](size_t i) const;
    const Node &operator[](const std::string &key) const;
    const std::vector<Node> &items() const { return items_; }
    const std::vector<std::string> &keys() const { return keys_; }

    // yaml-cpp style conversions: return false when the node is not a scalar that converts completely
    bool to(double &out) const;
    bool to(float &out) const;
    bool to(unsigned int &out) const;
    bool to(std::string &out) const;

private:
    friend class Parser;
    Type type_ = Undefined;
    Mark mark_;
    std::string text_;
    bool quoted_ = false;
    std::vector<Node> items_;        // sequence elements, or map values (parallel to keys_)
    std::vector<std::string> keys_;
};

// Throws ParseError on malformed input, std::ios_base::failure-like FileError when unreadable.
struct FileError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
Node load_file(const char *path);
Node load_string(const std::string &text);

} // namespace yamlsub
