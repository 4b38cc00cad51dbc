// yaml_subset.cpp -- see yaml_subset.h.
#include "yaml_subset.h"

#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace yamlsub {

static std::string format_error(const Mark &m, const std::string &msg)
{
    std::ostringstream s;
    s << "yaml-subset: error at line " << m.line + 1 << ", column " << m.column + 1 << ": " << msg;
    return s.str();
}

ParseError::ParseError(const Mark &m, const std::string &msg) : std::runtime_error(format_error(m, msg)), mark(m) {}

static const Node &undefined_node()
{
    static const Node n;
    return n;
}

const Node &Node::operator[](size_t i) const
{
    if (type_ == Sequence && i < items_.size()) return items_[i];
    return undefined_node();
}

const Node &Node::operator[](const std::string &key) const
{
    if (type_ == Map) {
        for (size_t i = 0; i < keys_.size(); i++)
            if (keys_[i] == key) return items_[i];
    }
    return undefined_node();
}

// ---- scalar conversions (yaml-cpp semantics: the whole token must convert) ---------------------------
static bool special_float(const std::string &t, double &out)
{
    static const char *pinf[] = {".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF"};
    static const char *ninf[] = {"-.inf", "-.Inf", "-.INF"};
    static const char *nan[] = {".nan", ".NaN", ".NAN"};
    for (const char *s : pinf)
        if (t == s) { out = std::numeric_limits<double>::infinity(); return true; }
    for (const char *s : ninf)
        if (t == s) { out = -std::numeric_limits<double>::infinity(); return true; }
    for (const char *s : nan)
        if (t == s) { out = std::numeric_limits<double>::quiet_NaN(); return true; }
    return false;
}

static bool plain_decimal(const std::string &t)
{
    if (t.empty()) return false;
    bool digit = false;
    for (char ch : t) {
        if (ch >= '0' && ch <= '9') digit = true;
        else if (!std::strchr("+-.eE", ch)) return false;
    }
    return digit;
}

bool Node::to(double &out) const
{
    if (type_ != Scalar) return false;
    if (special_float(text_, out)) return true;
    if (!plain_decimal(text_)) return false;
    errno = 0;
    char *end = nullptr;
    double v = std::strtod(text_.c_str(), &end);
    if (end != text_.c_str() + text_.size() || errno == ERANGE) return false;
    out = v;
    return true;
}

bool Node::to(float &out) const
{
    if (type_ != Scalar) return false;
    double d;
    if (special_float(text_, d)) { out = (float) d; return true; }
    if (!plain_decimal(text_)) return false;
    errno = 0;
    char *end = nullptr;
    float v = std::strtof(text_.c_str(), &end); // one rounding, like operator>>(float&)
    if (end != text_.c_str() + text_.size() || errno == ERANGE) return false;
    out = v;
    return true;
}

bool Node::to(unsigned int &out) const
{
    if (type_ != Scalar || text_.empty()) return false;
    if (text_[0] == '-') return false;
    for (char ch : text_)
        if (!std::isxdigit((unsigned char) ch) && ch != 'x' && ch != 'X' && ch != '+') return false;
    errno = 0;
    char *end = nullptr;
    unsigned long v = std::strtoul(text_.c_str(), &end, 0); // base from prefix, as a stream with basefield unset
    if (end != text_.c_str() + text_.size() || errno == ERANGE || v > UINT_MAX) return false;
    out = (unsigned int) v;
    return true;
}

bool Node::to(std::string &out) const
{
    if (type_ != Scalar) return false;
    out = text_;
    return true;
}

// ---- parser ------------------------------------------------------------------------------------------
struct Line {
    int no;
    int indent;
    std::string text;
};

static bool is_space(char c) { return c == ' ' || c == '\t'; }

static std::string rtrim(std::string s)
{
    while (!s.empty() && (is_space(s.back()) || s.back() == '\r')) s.pop_back();
    return s;
}

static size_t skip_spaces(const std::string &s, size_t pos)
{
    while (pos < s.size() && is_space(s[pos])) pos++;
    return pos;
}

// drop a trailing comment: '#' at the start or after blank, outside quotes
static std::string strip_comment(const std::string &s)
{
    char quote = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char ch = s[i];
        if (quote) {
            if (quote == '"' && ch == '\\') i++;
            else if (ch == quote) quote = 0;
        } else if (ch == '"' || ch == '\'') {
            // a quote only opens a quoted scalar at the start of a token
            if (i == 0 || is_space(s[i - 1]) || std::strchr("[{,:-", s[i - 1])) quote = ch;
        } else if (ch == '#' && (i == 0 || is_space(s[i - 1]))) {
            return s.substr(0, i);
        }
    }
    return s;
}

static bool is_seq_entry(const std::string &t) { return !t.empty() && t[0] == '-' && (t.size() == 1 || is_space(t[1])); }

// position of the ':' ending a block-mapping key in `t`, or npos
static size_t find_key_colon(const std::string &t)
{
    if (t.empty() || t[0] == '[' || t[0] == '{') return std::string::npos;
    size_t i = 0;
    if (t[0] == '"' || t[0] == '\'') {
        char q = t[0];
        for (i = 1; i < t.size(); i++) {
            if (q == '"' && t[i] == '\\') i++;
            else if (t[i] == q) break;
        }
        if (i >= t.size()) return std::string::npos;
        i = skip_spaces(t, i + 1);
        if (i < t.size() && t[i] == ':' && (i + 1 == t.size() || is_space(t[i + 1]))) return i;
        return std::string::npos;
    }
    for (; i < t.size(); i++) {
        if (t[i] == ':' && (i + 1 == t.size() || is_space(t[i + 1]))) return i;
    }
    return std::string::npos;
}

class Parser {
public:
    explicit Parser(const std::string &text) { split(text); }

    Node document()
    {
        if (lines_.empty()) {
            Node n;
            n.type_ = Node::Null;
            return n;
        }
        Node n = block(lines_[0].indent);
        if (cur_ < lines_.size()) fail(mark_of(lines_[cur_]), "unexpected content (bad indentation?)");
        return n;
    }

private:
    std::vector<Line> lines_;
    size_t cur_ = 0;

    [[noreturn]] static void fail(const Mark &m, const std::string &msg) { throw ParseError(m, msg); }
    static Mark mark_of(const Line &l) { return Mark{l.no, l.indent}; }

    void split(const std::string &text)
    {
        std::istringstream in(text);
        std::string raw;
        int no = 0;
        bool first_content = true;
        while (std::getline(in, raw)) {
            int lineno = no++;
            std::string s = rtrim(strip_comment(raw));
            size_t ind = 0;
            while (ind < s.size() && s[ind] == ' ') ind++;
            if (ind == s.size()) continue; // blank or comment only
            if (s[ind] == '\t') fail(Mark{lineno, (int) ind}, "tab characters are not allowed as indentation");
            std::string body = s.substr(ind);
            if (ind == 0 && body == "---") {
                if (!first_content) fail(Mark{lineno, 0}, "multiple documents are not supported");
                first_content = false;
                continue;
            }
            if (ind == 0 && body == "...") break;
            first_content = false;
            lines_.push_back(Line{lineno, (int) ind, body});
        }
    }

    Node block(int indent)
    {
        Line &l = lines_[cur_];
        if (l.indent != indent) fail(mark_of(l), "bad indentation");
        if (is_seq_entry(l.text)) return sequence(indent);
        if (find_key_colon(l.text) != std::string::npos) return mapping(indent);
        Mark m = mark_of(l);
        std::string t = l.text;
        cur_++;
        Node n = inline_value(t, m);
        if (cur_ < lines_.size() && lines_[cur_].indent > indent) fail(mark_of(lines_[cur_]), "multi-line scalars are not supported");
        return n;
    }

    Node mapping(int indent)
    {
        Node n;
        n.type_ = Node::Map;
        n.mark_ = mark_of(lines_[cur_]);
        while (cur_ < lines_.size() && lines_[cur_].indent == indent) {
            const Line l = lines_[cur_];
            if (is_seq_entry(l.text)) break;
            size_t colon = find_key_colon(l.text);
            if (colon == std::string::npos) fail(mark_of(l), "expected 'key: value'");
            Node keyn = scalar_node(rtrim(l.text.substr(0, colon)), mark_of(l));
            size_t vpos = skip_spaces(l.text, colon + 1);
            cur_++;
            Node value;
            if (vpos < l.text.size()) {
                value = inline_value(l.text.substr(vpos), Mark{l.no, indent + (int) vpos});
                if (cur_ < lines_.size() && lines_[cur_].indent > indent)
                    fail(mark_of(lines_[cur_]), "multi-line scalars are not supported");
            } else if (cur_ < lines_.size() && lines_[cur_].indent > indent) {
                value = block(lines_[cur_].indent);
            } else if (cur_ < lines_.size() && lines_[cur_].indent == indent && is_seq_entry(lines_[cur_].text)) {
                value = sequence(indent); // "key:\n- a\n- b": the dash may sit at the key's indentation
            } else {
                value.type_ = Node::Null;
                value.mark_ = Mark{l.no, indent + (int) colon + 1};
            }
            bool dup = false;
            for (const std::string &k : n.keys_) dup = dup || k == keyn.text_;
            if (!dup) { // first definition wins, as a lookup in yaml-cpp would find it
                n.keys_.push_back(keyn.text_);
                n.items_.push_back(std::move(value));
            }
        }
        if (cur_ < lines_.size() && lines_[cur_].indent > indent) fail(mark_of(lines_[cur_]), "bad indentation of a mapping entry");
        return n;
    }

    Node sequence(int indent)
    {
        Node n;
        n.type_ = Node::Sequence;
        n.mark_ = mark_of(lines_[cur_]);
        while (cur_ < lines_.size() && lines_[cur_].indent == indent && is_seq_entry(lines_[cur_].text)) {
            Line &l = lines_[cur_];
            size_t off = skip_spaces(l.text, 1);
            Node item;
            if (off >= l.text.size()) {
                Mark dash = mark_of(l);
                cur_++;
                if (cur_ < lines_.size() && lines_[cur_].indent > indent) item = block(lines_[cur_].indent);
                else { item.type_ = Node::Null; item.mark_ = dash; }
            } else {
                std::string rest = l.text.substr(off);
                int col = indent + (int) off;
                if (is_seq_entry(rest) || find_key_colon(rest) != std::string::npos) {
                    // "- key: value" / "- - x": the item is a block node that starts on this line
                    l.indent = col;
                    l.text = rest;
                    item = block(col);
                } else {
                    Mark m{l.no, col};
                    cur_++;
                    item = inline_value(rest, m);
                    if (cur_ < lines_.size() && lines_[cur_].indent > indent)
                        fail(mark_of(lines_[cur_]), "multi-line scalars are not supported");
                }
            }
            n.items_.push_back(std::move(item));
        }
        return n;
    }

    // a scalar or a flow collection written on the rest of a line
    Node inline_value(std::string text, const Mark &m)
    {
        text = rtrim(text);
        if (text.empty()) {
            Node n;
            n.type_ = Node::Null;
            n.mark_ = m;
            return n;
        }
        char c0 = text[0];
        if (c0 == '[' || c0 == '{') {
            // a flow collection may continue on following lines
            while (flow_depth(text) > 0 && cur_ < lines_.size()) text += " " + lines_[cur_++].text;
            size_t pos = 0;
            Node n = flow(text, pos, m, false);
            pos = skip_spaces(text, pos);
            if (pos != text.size()) fail(Mark{m.line, m.column + (int) pos}, "unexpected text after flow collection");
            return n;
        }
        if (c0 == '&' || c0 == '*' || c0 == '!' || c0 == '|' || c0 == '>' || c0 == '%' || c0 == '@' || c0 == '`')
            fail(m, std::string("unsupported YAML construct '") + c0 + "'");
        return scalar_node(text, m);
    }

    static int flow_depth(const std::string &s)
    {
        int depth = 0;
        char quote = 0;
        for (size_t i = 0; i < s.size(); i++) {
            char ch = s[i];
            if (quote) {
                if (quote == '"' && ch == '\\') i++;
                else if (ch == quote) quote = 0;
            } else if (ch == '"' || ch == '\'') quote = ch;
            else if (ch == '[' || ch == '{') depth++;
            else if (ch == ']' || ch == '}') depth--;
        }
        return depth;
    }

    static Node scalar_node(const std::string &raw, const Mark &m)
    {
        Node n;
        n.mark_ = m;
        if (!raw.empty() && (raw[0] == '"' || raw[0] == '\'')) {
            size_t pos = 0;
            n.text_ = quoted(raw, pos, m);
            if (skip_spaces(raw, pos) != raw.size()) fail(Mark{m.line, m.column + (int) pos}, "unexpected text after quoted scalar");
            n.type_ = Node::Scalar;
            n.quoted_ = true;
            return n;
        }
        if (raw.empty() || raw == "~" || raw == "null" || raw == "Null" || raw == "NULL") {
            n.type_ = Node::Null;
            return n;
        }
        n.type_ = Node::Scalar;
        n.text_ = raw;
        return n;
    }

    static std::string quoted(const std::string &s, size_t &pos, const Mark &m)
    {
        char q = s[pos++];
        std::string out;
        while (pos < s.size()) {
            char ch = s[pos++];
            if (q == '\'' && ch == '\'') {
                if (pos < s.size() && s[pos] == '\'') { out += '\''; pos++; continue; }
                return out;
            }
            if (q == '"' && ch == '"') return out;
            if (q == '"' && ch == '\\' && pos < s.size()) {
                char e = s[pos++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case '0': out += '\0'; break;
                default: out += e; break;
                }
                continue;
            }
            out += ch;
        }
        fail(m, "unterminated quoted scalar");
    }

    Node flow(const std::string &s, size_t &pos, const Mark &base, bool map_key)
    {
        pos = skip_spaces(s, pos);
        Mark m{base.line, base.column + (int) pos};
        if (pos >= s.size()) fail(m, "unexpected end of flow collection");
        Node n;
        n.mark_ = m;
        if (s[pos] == '[') {
            n.type_ = Node::Sequence;
            pos++;
            for (;;) {
                pos = skip_spaces(s, pos);
                if (pos >= s.size()) fail(m, "unterminated flow sequence");
                if (s[pos] == ']') { pos++; break; }
                n.items_.push_back(flow(s, pos, base, false));
                pos = skip_spaces(s, pos);
                if (pos < s.size() && s[pos] == ',') { pos++; continue; }
                if (pos < s.size() && s[pos] == ']') { pos++; break; }
                if (pos >= s.size()) fail(m, "unterminated flow sequence");
                fail(Mark{base.line, base.column + (int) pos}, "expected ',' or ']' in flow sequence");
            }
            return n;
        }
        if (s[pos] == '{') {
            n.type_ = Node::Map;
            pos++;
            for (;;) {
                pos = skip_spaces(s, pos);
                if (pos >= s.size()) fail(m, "unterminated flow mapping");
                if (s[pos] == '}') { pos++; break; }
                Node k = flow(s, pos, base, true);
                pos = skip_spaces(s, pos);
                if (pos >= s.size() || s[pos] != ':') fail(Mark{base.line, base.column + (int) pos}, "expected ':' in flow mapping");
                pos++;
                Node v = flow(s, pos, base, false);
                n.keys_.push_back(k.text_);
                n.items_.push_back(std::move(v));
                pos = skip_spaces(s, pos);
                if (pos < s.size() && s[pos] == ',') { pos++; continue; }
                if (pos < s.size() && s[pos] == '}') { pos++; break; }
                if (pos >= s.size()) fail(m, "unterminated flow mapping");
                fail(Mark{base.line, base.column + (int) pos}, "expected ',' or '}' in flow mapping");
            }
            return n;
        }
        if (s[pos] == '"' || s[pos] == '\'') {
            n.text_ = quoted(s, pos, m);
            n.type_ = Node::Scalar;
            n.quoted_ = true;
            return n;
        }
        size_t start = pos;
        while (pos < s.size()) {
            char ch = s[pos];
            if (ch == ',' || ch == ']' || ch == '}') break;
            if (map_key && ch == ':' && (pos + 1 == s.size() || is_space(s[pos + 1]))) break;
            pos++;
        }
        Node sc = scalar_node(rtrim(s.substr(start, pos - start)), m);
        return sc;
    }
};

Node load_string(const std::string &text)
{
    Parser p(text);
    return p.document();
}

Node load_file(const char *path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw FileError(std::string("cannot open ") + path);
    std::ostringstream ss;
    ss << in.rdbuf();
    return load_string(ss.str());
}

} // namespace yamlsub
