// Loaders of the host-side API: Wavefront OBJ (src/obj.rs), YAML scene descriptions
// (src/bin/yaml_renderer.rs) and the scene-graph text format (.sg).
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>

#include "rt_scene.hpp"

namespace rt {

static std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static std::vector<std::string> split(const std::string& s, char sep) {  // like str::split: keeps empty pieces
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        size_t p = s.find(sep, start);
        if (p == std::string::npos) {
            out.push_back(s.substr(start));
            break;
        }
        out.push_back(s.substr(start, p - start));
        start = p + 1;
    }
    return out;
}
static double parse_f64(const std::string& s) {
    if (s.empty()) throw Error("cannot parse float from an empty string");
    errno = 0;
    char* end = nullptr;
    double v = std::strtod(s.c_str(), &end);
    if (end != s.c_str() + s.size()) throw Error("invalid float literal: " + s);
    return v;
}
static size_t parse_usize(const std::string& s) {
    if (s.empty()) throw Error("cannot parse integer from an empty string");
    size_t k = s[0] == '+' ? 1 : 0;
    if (k >= s.size()) throw Error("invalid digit found in string: " + s);
    size_t v = 0;
    for (; k < s.size(); k++) {
        if (s[k] < '0' || s[k] > '9') throw Error("invalid digit found in string: " + s);
        v = v * 10 + (size_t)(s[k] - '0');
    }
    return v;
}

// ------------------------------------------------------------------ OBJ (src/obj.rs:16-135)
ObjParser ObjParser::from_file(const std::string& path) { return from_str(read_file(path)); }

// 1-based index lists with a dummy entry 0; `idxs[k]` = (vertex index, normal index)
std::vector<Triangle> ObjParser::fan_triangulation(const std::vector<std::pair<size_t, size_t>>& idxs) const {
    std::vector<Triangle> out;
    auto vtx = [&](size_t i) {
        if (i >= vertices.size()) throw Error("obj: vertex index out of range");
        return vertices[i];
    };
    auto nrm = [&](size_t i) {
        if (i >= normals.size()) throw Error("obj: normal index out of range");
        return normals[i];
    };
    if (idxs.size() < 2) throw Error("obj: face without vertices");
    Tuple p1 = vtx(idxs[1].first), n1 = nrm(idxs[1].second);
    for (size_t k = 2; k + 1 < idxs.size(); k++) {
        Tuple p2 = vtx(idxs[k].first), n2 = nrm(idxs[k].second);
        Tuple p3 = vtx(idxs[k + 1].first), n3 = nrm(idxs[k + 1].second);
        if (idxs[1].second != 0)  // smooth iff the FIRST vertex carries a normal index (:72)
            out.push_back(Triangle::smooth(p1, p2, p3, n1, n2, n3));
        else
            out.push_back(Triangle(p1, p2, p3));
    }
    return out;
}

ObjParser ObjParser::from_str(const std::string& contents) {
    ObjParser ps;
    ps.vertices.push_back(pt(0, 0, 0));
    ps.normals.push_back(pt(0, 0, 0));
    int current = -1;  // index into groups, -1 = default group
    for (const std::string& raw : split(contents, '\n')) {
        std::string line = trim(raw);
        if (line.empty()) continue;
        std::vector<std::string> ex = split(line, ' ');
        const std::string& cmd = ex[0];
        auto floats = [&]() {  // parse_f64s, :48-60: pieces are trimmed, empty ones skipped
            std::vector<double> f{0.0};
            for (size_t k = 1; k < ex.size(); k++) {
                std::string p = trim(ex[k]);
                if (p.empty()) continue;
                f.push_back(parse_f64(p));
            }
            if (f.size() < 4) throw Error("obj: fewer than three coordinates in: " + line);
            return f;
        };
        if (cmd == "v") {
            auto f = floats();
            ps.vertices.push_back(pt(f[1], f[2], f[3]));
        } else if (cmd == "vn") {
            auto f = floats();
            ps.normals.push_back(pt(f[1], f[2], f[3]));  // normals are stored as points (w = 1), :114
        } else if (cmd == "f") {
            std::vector<std::pair<size_t, size_t>> idx{{0, 0}};
            for (size_t k = 1; k < ex.size(); k++) {  // parse_faces, :30-46: empty pieces are an error
                std::vector<std::string> is = split(ex[k], '/');
                size_t vi = parse_usize(is[0]);
                size_t ni = is.size() >= 3 ? parse_usize(is[2]) : 0;
                idx.push_back({vi, ni});
            }
            for (Triangle& t : ps.fan_triangulation(idx)) {
                Group& g = current < 0 ? ps.default_group : ps.groups[current].second;
                g.add_child(t.shape());
            }
        } else if (cmd == "g") {
            if (ex.size() < 2) throw Error("obj: group statement without a name");
            const std::string& name = ex[1];
            current = -1;
            for (size_t k = 0; k < ps.groups.size(); k++)
                if (ps.groups[k].first == name) current = (int)k;
            if (current < 0) {
                ps.groups.emplace_back(name, Group());
                current = (int)ps.groups.size() - 1;
            } else {
                ps.groups[current].second = Group();  // HashMap::insert replaces an existing entry
            }
        } else {
            ps.lines_ignored++;
        }
    }
    return ps;
}

// The reference drains a HashMap here, so the order of several named groups is unspecified
// (src/obj.rs:131-133); this loader keeps file order.
Group ObjParser::into_group() {
    Group g;
    g.add_child(default_group.shape());
    for (auto& kv : groups) g.add_child(kv.second.shape());
    groups.clear();
    return g;
}

// ------------------------------------------------------------------ YAML subset
namespace {

struct YNode {
    enum Type { Null, Int, Real, Bool, Str, Seq, Map } type = Null;
    std::string s;  // scalar text
    double num = 0;
    bool b = false;
    std::vector<YNode> seq;
    std::vector<std::pair<std::string, YNode>> map;
    const YNode& operator[](const std::string& key) const {
        static const YNode bad;
        if (type != Map) return bad;
        for (auto& kv : map)
            if (kv.first == key) return kv.second;
        return bad;
    }
    const YNode& operator[](size_t i) const {
        static const YNode bad;
        if (type != Seq || i >= seq.size()) return bad;
        return seq[i];
    }
    bool is_str() const { return type == Str; }
};

YNode scalar_node(const std::string& text) {
    YNode n;
    std::string t = trim(text);
    n.s = t;
    if (t.empty() || t == "~" || t == "null") {
        n.type = YNode::Null;
        return n;
    }
    if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\''))) {
        n.type = YNode::Str;
        n.s = t.substr(1, t.size() - 2);
        return n;
    }
    if (t == "true" || t == "false") {
        n.type = YNode::Bool;
        n.b = t == "true";
        return n;
    }
    {  // integer: [+-]?digits
        size_t k = (t[0] == '+' || t[0] == '-') ? 1 : 0;
        bool digits = k < t.size();
        for (size_t q = k; q < t.size(); q++) digits = digits && std::isdigit((unsigned char)t[q]);
        if (digits) {
            n.type = YNode::Int;
            n.num = (double)std::strtoll(t.c_str(), nullptr, 10);
            return n;
        }
    }
    if (t == ".inf" || t == "+.inf" || t == ".Inf" || t == ".INF") {
        n.type = YNode::Real, n.num = HUGE_VAL;
        return n;
    }
    if (t == "-.inf" || t == "-.Inf" || t == "-.INF") {
        n.type = YNode::Real, n.num = -HUGE_VAL;
        return n;
    }
    {
        char* end = nullptr;
        double v = std::strtod(t.c_str(), &end);
        bool numeric_start = std::isdigit((unsigned char)t[0]) || t[0] == '-' || t[0] == '+' || t[0] == '.';
        if (numeric_start && end == t.c_str() + t.size()) {
            n.type = YNode::Real, n.num = v;
            return n;
        }
    }
    n.type = YNode::Str;
    return n;
}

struct Line {
    int indent;
    std::string text;  // without indentation and trailing comment
    int number;
};

std::string strip_comment(const std::string& s) {
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (c == '\'' && !in_d) in_s = !in_s;
        if (c == '"' && !in_s) in_d = !in_d;
        if (c == '#' && !in_s && !in_d && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
    }
    return s;
}

struct YParser {
    std::vector<Line> lines;
    size_t pos = 0;

    explicit YParser(const std::string& text) {
        int number = 0;
        for (const std::string& raw : split(text, '\n')) {
            number++;
            std::string s = strip_comment(raw);
            while (!s.empty() && (s.back() == '\r' || std::isspace((unsigned char)s.back()))) s.pop_back();
            size_t ind = 0;
            while (ind < s.size() && s[ind] == ' ') ind++;
            if (ind < s.size() && s[ind] == '\t') throw Error("yaml: tab indentation at line " + std::to_string(number));
            if (ind == s.size()) continue;
            if (s.compare(ind, 3, "---") == 0 && ind == 0) continue;
            lines.push_back({(int)ind, s.substr(ind), number});
        }
    }
    [[noreturn]] void fail(const std::string& msg, int line) { throw Error("yaml: " + msg + " at line " + std::to_string(line)); }

    // flow sequence "[a, b, [c, d]]"
    YNode parse_flow(const std::string& t, size_t& i, int line) {
        YNode n;
        n.type = YNode::Seq;
        i++;  // '['
        std::string cur;
        bool have = false;
        for (;;) {
            if (i >= t.size()) fail("unterminated flow sequence", line);
            char c = t[i];
            if (c == '[') {
                n.seq.push_back(parse_flow(t, i, line));
                have = false;
                cur.clear();
                while (i < t.size() && std::isspace((unsigned char)t[i])) i++;
                if (i < t.size() && t[i] == ',') i++;
                else if (i < t.size() && t[i] == ']') { i++; return n; }
                continue;
            }
            if (c == ',' || c == ']') {
                if (have || !trim(cur).empty()) n.seq.push_back(scalar_node(cur));
                cur.clear();
                have = false;
                i++;
                if (c == ']') return n;
                continue;
            }
            cur += c;
            if (!std::isspace((unsigned char)c)) have = true;
            i++;
        }
    }
    YNode parse_inline(const std::string& text, int line) {
        std::string t = trim(text);
        if (!t.empty() && t[0] == '[') {
            size_t i = 0;
            YNode n = parse_flow(t, i, line);
            if (!trim(t.substr(i)).empty()) fail("trailing characters after flow sequence", line);
            return n;
        }
        if (!t.empty() && t[0] == '{') fail("flow mappings are not supported", line);
        return scalar_node(t);
    }
    // "key: value" split; returns false if the text is not a mapping entry
    static bool split_key(const std::string& t, std::string* key, std::string* rest) {
        if (t.empty() || t[0] == '[' || t[0] == '"' || t[0] == '\'') return false;
        for (size_t i = 0; i < t.size(); i++) {
            if (t[i] == ':' && (i + 1 == t.size() || t[i + 1] == ' ')) {
                *key = trim(t.substr(0, i));
                *rest = i + 1 < t.size() ? trim(t.substr(i + 1)) : "";
                return !key->empty();
            }
        }
        return false;
    }

    // parses the block that starts at lines[pos] and whose lines have indentation >= indent
    YNode parse_block(int indent) {
        if (pos >= lines.size()) return YNode();
        const Line& first = lines[pos];
        if (first.indent < indent) return YNode();
        int bi = first.indent;
        if (first.text[0] == '-' && (first.text.size() == 1 || first.text[1] == ' ')) return parse_seq(bi);
        std::string k, r;
        if (split_key(first.text, &k, &r)) return parse_map(bi);
        pos++;
        return parse_inline(first.text, first.number);
    }
    YNode parse_seq(int bi) {
        YNode n;
        n.type = YNode::Seq;
        while (pos < lines.size() && lines[pos].indent == bi && lines[pos].text[0] == '-' &&
               (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            Line& ln = lines[pos];
            size_t off = 1;
            while (off < ln.text.size() && ln.text[off] == ' ') off++;
            if (off >= ln.text.size()) {  // "-" alone: nested block on following lines
                pos++;
                n.seq.push_back(parse_block(bi + 1));
                continue;
            }
            // re-interpret the remainder as a line of its own, indented to its column
            ln.indent = bi + (int)off;
            ln.text = ln.text.substr(off);
            n.seq.push_back(parse_block(ln.indent));
        }
        if (pos < lines.size() && lines[pos].indent > bi) fail("unexpected indentation", lines[pos].number);
        return n;
    }
    YNode parse_map(int bi) {
        YNode n;
        n.type = YNode::Map;
        while (pos < lines.size() && lines[pos].indent == bi) {
            const Line& ln = lines[pos];
            std::string key, rest;
            if (ln.text[0] == '-' && (ln.text.size() == 1 || ln.text[1] == ' ')) break;
            if (!split_key(ln.text, &key, &rest)) fail("expected 'key: value'", ln.number);
            pos++;
            YNode value;
            if (!rest.empty()) {
                value = parse_inline(rest, ln.number);
            } else if (pos < lines.size()) {
                const Line& nx = lines[pos];
                bool seq_same_indent = nx.indent == bi && nx.text[0] == '-' && (nx.text.size() == 1 || nx.text[1] == ' ');
                if (nx.indent > bi)
                    value = parse_block(nx.indent);
                else if (seq_same_indent)
                    value = parse_seq(bi);
            }
            n.map.emplace_back(key, std::move(value));
        }
        if (pos < lines.size() && lines[pos].indent > bi) fail("unexpected indentation", lines[pos].number);
        return n;
    }
    YNode parse_document() {
        if (lines.empty()) return YNode();
        YNode n = parse_block(0);
        if (pos < lines.size()) fail("unexpected content", lines[pos].number);
        return n;
    }
};

// ---- scene construction from the YAML tree (src/bin/yaml_renderer.rs:6-357)
struct Definitions {
    std::map<std::string, Matrix4x4> transforms;
    std::map<std::string, Material> materials;
    std::map<std::string, YNode> shapes;
};

double as_float(const YNode& n) {  // :69-77
    if (n.type == YNode::Real || n.type == YNode::Int) return n.num;
    throw Error("Unable to read f64");
}
const std::string& as_str(const YNode& n, const char* what) {
    if (!n.is_str()) throw Error(std::string(what));
    return n.s;
}
Tuple as_v(const YNode& n) { return v(as_float(n[0]), as_float(n[1]), as_float(n[2])); }
Tuple as_pt(const YNode& n) { return pt(as_float(n[0]), as_float(n[1]), as_float(n[2])); }
Color as_color(const YNode& n) { return {as_float(n[0]), as_float(n[1]), as_float(n[2])}; }

Matrix4x4 populate_transform(const YNode& p, Matrix4x4 t) {  // :168-200
    const std::string& type = as_str(p[0], "tranform type is not a String");
    if (type == "rotate-x") return t.rotate_x(as_float(p[1]));
    if (type == "rotate-y") return t.rotate_y(as_float(p[1]));
    if (type == "rotate-z") return t.rotate_z(as_float(p[1]));
    if (type == "scale") return t.scale(as_float(p[1]), as_float(p[2]), as_float(p[3]));
    if (type == "translate") return t.translate(as_float(p[1]), as_float(p[2]), as_float(p[3]));
    throw Error("Unrecognized plane transform: " + type);
}
template <class M>
const typename M::mapped_type& lookup(const M& m, const std::string& key, const char* what) {
    auto it = m.find(key);
    if (it == m.end()) throw Error(std::string("undefined ") + what + ": " + key);
    return it->second;
}
Matrix4x4 as_transform(const YNode& n, const Definitions& defs) {  // :156-166
    if (n.type != YNode::Seq) throw Error("transforms should be an array");
    Matrix4x4 t = id();
    for (const YNode& p : n.seq) {
        if (p.type == YNode::Seq)
            t = populate_transform(p, t);
        else if (p.is_str())
            t = lookup(defs.transforms, p.s, "transform");  // a name REPLACES what was accumulated (:161)
        else
            throw Error("transform list entry is neither an array nor a name");
    }
    return t;
}
void pattern_props(const YNode& n, const Definitions& defs, Color* a, Color* b, Matrix4x4* t) {  // :335-356
    *a = Color{}, *b = Color{}, *t = id();
    for (auto& kv : n.map) {
        if (kv.first == "type") {
        } else if (kv.first == "colors") {
            *a = as_color(kv.second[0]);
            *b = as_color(kv.second[1]);
        } else if (kv.first == "transform")
            *t = as_transform(kv.second, defs);
        else
            throw Error("Unexpected pattern property: " + kv.first);
    }
}
Material populate_material(const YNode& n, Material mat, const Definitions& defs) {  // :206-251
    if (n.type != YNode::Map) throw Error("material properties are not a Hash");
    for (auto& kv : n.map) {
        const std::string& key = kv.first;
        const YNode& val = kv.second;
        if (key == "pattern") {
            const std::string& type = as_str(val["type"], "pattern type is not a String");
            Color a, b;
            Matrix4x4 t;
            if (type == "checkers") {
                pattern_props(val, defs, &a, &b, &t);
                Pattern p = checkers_pattern(a, b);
                p.transform = t;
                mat.set_pattern(p);
            } else if (type == "stripes") {
                pattern_props(val, defs, &a, &b, &t);
                Pattern p = stripe_pattern(a, b);
                p.transform = t;
                mat.set_pattern(p);
            } else
                throw Error("Unexpected pattern type: " + type);
        } else if (key == "color")
            mat.color = as_color(val);
        else if (key == "ambient")
            mat.ambient = as_float(val);
        else if (key == "diffuse")
            mat.diffuse = as_float(val);
        else if (key == "specular")
            mat.specular = as_float(val);
        else if (key == "reflective")
            mat.reflective = as_float(val);
        else if (key == "shininess")
            mat.shininess = as_float(val);
        else if (key == "transparency")
            mat.transparency = as_float(val);
        else if (key == "refractive-index")
            mat.refractive_index = as_float(val);
        else
            throw Error("Unknown material property: " + key);
    }
    return mat;
}

ShapePtr as_shape(const YNode& n, const Definitions& defs);

ShapePtr populate_shape(const YNode& n, ShapePtr shape, const Definitions& defs) {  // :317-333
    const YNode& t = n["transform"];
    if (t.type == YNode::Seq)
        shape->set_transform(as_transform(t, defs));
    else if (t.is_str())
        shape->set_transform(lookup(defs.transforms, t.s, "transform"));
    const YNode& mt = n["material"];
    if (mt.type == YNode::Map)
        shape->set_material(populate_material(mt, Material(), defs));
    else if (mt.is_str())
        shape->set_material(lookup(defs.materials, mt.s, "material"));
    if (n["shadow"].type == YNode::Bool) shape->set_shadow(n["shadow"].b);
    return shape;
}
ShapePtr as_shape(const YNode& n, const Definitions& defs) {  // :253-315
    const std::string& type = as_str(n["add"], "add parameter is not a String");
    if (type == "cube") return populate_shape(n, Cube().shape(), defs);
    if (type == "plane") return populate_shape(n, Plane().shape(), defs);
    if (type == "sphere") return populate_shape(n, Sphere().shape(), defs);
    if (type == "cylinder" || type == "cone") {
        double mn = as_float(n["min"]), mx = as_float(n["max"]);
        if (n["closed"].type != YNode::Bool) throw Error("closed property is not a Bool");
        bool closed = n["closed"].b;
        ShapePtr s = type == "cone" ? Cone(mn, mx, closed).shape() : Cylinder(mn, mx, closed).shape();
        return populate_shape(n, std::move(s), defs);
    }
    if (type == "group") {
        if (n["children"].type != YNode::Seq) throw Error("children property not an Array");
        Group g;
        for (const YNode& ch : n["children"].seq) g.add_child(as_shape(ch, defs));
        return populate_shape(n, g.shape(), defs);
    }
    auto it = defs.shapes.find(type);
    if (it == defs.shapes.end()) throw Error("Unknown shape: " + type);
    return populate_shape(n, as_shape(it->second, defs), defs);
}

void define(const YNode& obj, Definitions& defs) {  // :21-49 -- routing is by SUBSTRING of the name
    const std::string& name = as_str(obj["define"], "define is not a String");
    auto has = [&](const char* s) { return name.find(s) != std::string::npos; };
    if (has("leg") || has("cap") || has("wacky")) {
        defs.shapes[name] = obj["value"];
    } else if (has("-material")) {
        Material base;
        if (obj["extend"].is_str()) base = lookup(defs.materials, obj["extend"].s, "material");
        defs.materials[name] = populate_material(obj["value"], base, defs);
    } else if (has("-transform")) {
        defs.transforms[name] = as_transform(obj["value"], defs);
    } else if (has("-object")) {
        const YNode& val = obj["value"];
        if (val.type == YNode::Seq) {  // [base-name, one-op]: only value[1] is applied (:39-44)
            const std::string& base = as_str(val[0], "base object name not a String");
            Matrix4x4 t = lookup(defs.transforms, base, "transform");
            defs.transforms[name] = populate_transform(val[1], t);
        }
    } else
        throw Error("Unexpected define type: " + name + "!");
}

std::unique_ptr<Camera> as_camera(const YNode& n) {  // :103-137
    bool hw = false, hh = false, hf = false, hfrom = false, hto = false, hup = false;
    size_t width = 0, height = 0;
    double fov = 0;
    Tuple from, to, up;
    for (auto& kv : n.map) {
        const std::string& key = kv.first;
        if (key == "add") {
        } else if (key == "width")
            width = (size_t)as_float(kv.second), hw = true;
        else if (key == "height")
            height = (size_t)as_float(kv.second), hh = true;
        else if (key == "field-of-view")
            fov = as_float(kv.second), hf = true;
        else if (key == "from")
            from = as_v(kv.second), hfrom = true;
        else if (key == "to")
            to = as_v(kv.second), hto = true;
        else if (key == "up")
            up = as_v(kv.second), hup = true;
        else
            throw Error("Unexpected camera property: " + key);
    }
    if (!hw) throw Error("camera missing required field 'width'");
    if (!hh) throw Error("camera missing required field 'height'");
    if (!hf) throw Error("camera missing required field 'field-of-view'");
    if (!hfrom) throw Error("camera missing required field 'from'");
    if (!hto) throw Error("camera missing required field 'to'");
    if (!hup) throw Error("camera missing required field 'up'");
    auto cam = std::make_unique<Camera>(width, height, fov);
    cam->transform = view_transform(from, to, up);
    return cam;
}
PointLight as_light(const YNode& n) {  // :139-154
    PointLight l;
    for (auto& kv : n.map) {
        if (kv.first == "add") {
        } else if (kv.first == "at")
            l.position = as_pt(kv.second);
        else if (kv.first == "intensity")
            l.intensity = as_color(kv.second);
        else
            throw Error("Unexpected light property: " + kv.first);
    }
    return l;
}

}  // namespace

std::unique_ptr<Scene> load_yaml_scene_str(const std::string& text) {  // YamlScene::parse, :377-408
    YParser yp(text);
    YNode doc = yp.parse_document();
    if (doc.type != YNode::Seq) throw Error("yaml: the scene must be a list of add:/define: entries");
    auto sc = std::make_unique<Scene>();
    Definitions defs;
    for (const YNode& obj : doc.seq) {
        if (obj["add"].is_str()) {
            const std::string& type = obj["add"].s;
            if (type == "camera")
                sc->camera = as_camera(obj);
            else if (type == "light")
                sc->world.lights.push_back(as_light(obj));
            else
                sc->world.objects.push_back(as_shape(obj, defs));
        } else if (obj["define"].is_str()) {
            define(obj, defs);
        } else
            throw Error("Unexpected object type in scene list");
    }
    return sc;
}
std::unique_ptr<Scene> load_yaml_scene(const std::string& path) { return load_yaml_scene_str(read_file(path)); }

// ------------------------------------------------------------------ .sg
namespace {
std::string num(double v) {
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    if (std::isnan(v)) return "nan";
    char buf[40];
    for (int prec = 1; prec <= 17; prec++) {
        std::snprintf(buf, sizeof buf, "%.*g", prec, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    return buf;
}
struct SgWriter {
    std::ostringstream defs, tree;
    std::vector<Matrix4x4> xfs{Matrix4x4::identity()};
    std::vector<Material> mats;
    int xf_id(const Matrix4x4& m) {
        for (size_t i = 0; i < xfs.size(); i++)
            if (std::memcmp(&xfs[i], &m, sizeof m) == 0) return (int)i;
        xfs.push_back(m);
        defs << "xf " << xfs.size() - 1;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) defs << ' ' << num(m.m[r][c]);
        defs << '\n';
        return (int)xfs.size() - 1;
    }
    static bool same(const Material& a, const Material& b) {
        auto bits = [](const void* p, const void* q, size_t n) { return std::memcmp(p, q, n) == 0; };
        if (!bits(&a.color, &b.color, sizeof a.color) || a.has_pattern != b.has_pattern) return false;
        if (a.has_pattern && !(a.pattern.design == b.pattern.design && bits(&a.pattern.a, &b.pattern.a, sizeof(Color)) &&
                               bits(&a.pattern.b, &b.pattern.b, sizeof(Color)) &&
                               bits(&a.pattern.transform, &b.pattern.transform, sizeof(Matrix4x4))))
            return false;
        return bits(&a.ambient, &b.ambient, 7 * sizeof(double));
    }
    int mat_id(const Material& m) {
        for (size_t i = 0; i < mats.size(); i++)
            if (same(mats[i], m)) return (int)i;
        int pxf = m.has_pattern ? xf_id(m.pattern.transform) : 0;
        mats.push_back(m);
        defs << "mat " << mats.size() - 1 << ' ' << num(m.color.red) << ' ' << num(m.color.green) << ' ' << num(m.color.blue)
             << ' ' << num(m.ambient) << ' ' << num(m.diffuse) << ' ' << num(m.specular) << ' ' << num(m.shininess) << ' '
             << num(m.reflective) << ' ' << num(m.transparency) << ' ' << num(m.refractive_index) << ' ';
        if (!m.has_pattern)
            defs << "none";
        else {
            static const char* names[] = {"stripe", "gradient", "ring", "checkers", "test"};
            defs << names[(int)m.pattern.design] << ' ' << num(m.pattern.a.red) << ' ' << num(m.pattern.a.green) << ' '
                 << num(m.pattern.a.blue) << ' ' << num(m.pattern.b.red) << ' ' << num(m.pattern.b.green) << ' '
                 << num(m.pattern.b.blue) << ' ' << pxf;
        }
        defs << '\n';
        return (int)mats.size() - 1;
    }
    void p3(const Tuple& t) { tree << ' ' << num(t.x) << ' ' << num(t.y) << ' ' << num(t.z); }
    void t4(const Tuple& t) { tree << ' ' << num(t.x) << ' ' << num(t.y) << ' ' << num(t.z) << ' ' << num(t.w); }
    void node(const Shape& s) {
        int mid = mat_id(s.material()), xid = xf_id(s.transform());
        auto head = [&](const char* kind) {
            tree << "node " << kind << " mat " << mid << " xf " << xid << " shadow " << (s.shadow() ? 1 : 0);
        };
        switch (s.kind()) {
            case Kind::Sphere: head("sphere"); tree << '\n'; break;
            case Kind::Plane: head("plane"); tree << '\n'; break;
            case Kind::Cube: head("cube"); tree << '\n'; break;
            case Kind::Cylinder: {
                auto& c = static_cast<const Cylinder&>(s);
                head("cylinder");
                tree << ' ' << num(c.min) << ' ' << num(c.max) << ' ' << (c.closed ? 1 : 0) << '\n';
                break;
            }
            case Kind::Cone: {
                auto& c = static_cast<const Cone&>(s);
                head("cone");
                tree << ' ' << num(c.min) << ' ' << num(c.max) << ' ' << (c.closed ? 1 : 0) << '\n';
                break;
            }
            case Kind::Triangle: {
                auto& t = static_cast<const Triangle&>(s);
                head(t.is_smooth ? "stri" : "tri");
                p3(t.p1), p3(t.p2), p3(t.p3);
                if (t.is_smooth) t4(t.n1), t4(t.n2), t4(t.n3);
                tree << '\n';
                break;
            }
            case Kind::Group: {
                auto& g = static_cast<const Group&>(s);
                head("group");
                tree << ' ' << g.children.size() << '\n';
                for (auto& ch : g.children) node(*ch);
                break;
            }
            case Kind::Csg: {
                auto& c = static_cast<const Csg&>(s);
                head("csg");
                tree << ' ' << (c.op == Op::Union ? "union" : c.op == Op::Intersection ? "intersection" : "difference") << '\n';
                node(*c.left);
                node(*c.right);
                break;
            }
        }
    }
};
}  // namespace

void save_sg(const Scene& scene, const std::string& path) {
    if (!scene.camera) throw Error("sg: the scene has no camera");
    SgWriter w;
    int cam_xf = w.xf_id(scene.camera->transform);
    for (auto& o : scene.world.objects) w.node(*o);
    std::ofstream f(path);
    if (!f) throw Error("cannot write " + path);
    f << "rtsg 1\n" << w.defs.str();
    f << "camera " << scene.camera->hsize << ' ' << scene.camera->vsize << ' ' << num(scene.camera->field_of_view) << ' '
      << cam_xf << '\n';
    for (auto& l : scene.world.lights)
        f << "light " << num(l.position.x) << ' ' << num(l.position.y) << ' ' << num(l.position.z) << ' '
          << num(l.intensity.red) << ' ' << num(l.intensity.green) << ' ' << num(l.intensity.blue) << '\n';
    f << "objects " << scene.world.objects.size() << '\n' << w.tree.str() << "end\n";
}

namespace {
struct SgReader {
    std::ifstream in;
    std::vector<std::string> tk;
    size_t ti = 0;
    std::map<int, Matrix4x4> xfs;
    std::map<int, Material> mats;
    bool next_line() {
        std::string line;
        while (std::getline(in, line)) {
            size_t h = line.find('#');
            if (h != std::string::npos) line.resize(h);
            std::istringstream ss(line);
            tk.clear();
            ti = 0;
            std::string w;
            while (ss >> w) tk.push_back(w);
            if (!tk.empty()) return true;
        }
        return false;
    }
    const std::string& word() {
        if (ti >= tk.size()) throw Error("sg: truncated record");
        return tk[ti++];
    }
    double number() { return std::strtod(word().c_str(), nullptr); }
    int integer() { return std::atoi(word().c_str()); }
    void expect(const char* s) {
        if (word() != s) throw Error(std::string("sg: expected ") + s);
    }
    ShapePtr node() {
        if (!next_line()) throw Error("sg: missing node");
        expect("node");
        std::string kind = word();
        expect("mat");
        int mid = integer();
        expect("xf");
        int xid = integer();
        expect("shadow");
        bool shadow = integer() != 0;
        auto p3 = [&]() {
            double x = number(), y = number(), z = number();
            return pt(x, y, z);
        };
        auto t4 = [&]() {
            double x = number(), y = number(), z = number(), w = number();
            return Tuple{x, y, z, w};
        };
        ShapePtr s;
        if (kind == "sphere")
            s = Sphere().shape();
        else if (kind == "plane")
            s = Plane().shape();
        else if (kind == "cube")
            s = Cube().shape();
        else if (kind == "cylinder" || kind == "cone") {
            double mn = number(), mx = number();
            bool closed = integer() != 0;
            s = kind == "cone" ? Cone(mn, mx, closed).shape() : Cylinder(mn, mx, closed).shape();
        } else if (kind == "tri") {
            Tuple a = p3(), b = p3(), c = p3();
            s = Triangle(a, b, c).shape();
        } else if (kind == "stri") {
            Tuple a = p3(), b = p3(), c = p3();
            Tuple n1 = t4(), n2 = t4(), n3 = t4();
            s = Triangle::smooth(a, b, c, n1, n2, n3).shape();
        } else if (kind == "group") {
            int n = integer();
            Group g;
            for (int k = 0; k < n; k++) g.add_child(node());
            s = g.shape();
        } else if (kind == "csg") {
            std::string op = word();
            Op o = op == "union" ? Op::Union : op == "intersection" ? Op::Intersection : Op::Difference;
            if (op != "union" && op != "intersection" && op != "difference") throw Error("sg: bad csg op");
            ShapePtr l = node();
            ShapePtr r = node();
            s = Csg(o, std::move(l), std::move(r)).shape();
        } else
            throw Error("sg: unknown node kind " + kind);
        s->set_material(lookup(mats, std::to_string(mid)));
        s->set_transform(lookup(xfs, std::to_string(xid)));
        s->set_shadow(shadow);
        return s;
    }
    template <class T>
    const T& lookup(const std::map<int, T>& m, const std::string& id) {
        auto it = m.find(std::atoi(id.c_str()));
        if (it == m.end()) throw Error("sg: undefined id " + id);
        return it->second;
    }
};
}  // namespace

std::unique_ptr<Scene> load_sg(const std::string& path) {
    SgReader rd;
    rd.in.open(path);
    if (!rd.in) throw Error("cannot open " + path);
    rd.xfs[0] = Matrix4x4::identity();
    if (!rd.next_line() || rd.word() != "rtsg") throw Error("sg: bad magic in " + path);
    auto sc = std::make_unique<Scene>();
    while (rd.next_line()) {
        std::string cmd = rd.word();
        if (cmd == "xf") {
            int idn = rd.integer();
            Matrix4x4 m;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) m.m[r][c] = rd.number();
            rd.xfs[idn] = m;
        } else if (cmd == "mat") {
            int idn = rd.integer();
            Material mt;
            mt.color.red = rd.number(), mt.color.green = rd.number(), mt.color.blue = rd.number();
            mt.ambient = rd.number(), mt.diffuse = rd.number(), mt.specular = rd.number(), mt.shininess = rd.number();
            mt.reflective = rd.number(), mt.transparency = rd.number(), mt.refractive_index = rd.number();
            std::string pk = rd.word();
            static const char* names[] = {"stripe", "gradient", "ring", "checkers", "test"};
            if (pk != "none") {
                int kind = -1;
                for (int k = 0; k < 5; k++)
                    if (pk == names[k]) kind = k;
                if (kind < 0) throw Error("sg: bad pattern kind " + pk);
                Pattern p;
                p.design = (PatternDesign)kind;
                p.a.red = rd.number(), p.a.green = rd.number(), p.a.blue = rd.number();
                p.b.red = rd.number(), p.b.green = rd.number(), p.b.blue = rd.number();
                p.transform = rd.lookup(rd.xfs, rd.word());
                mt.set_pattern(p);
            }
            rd.mats[idn] = mt;
        } else if (cmd == "camera") {
            size_t h = (size_t)rd.integer(), vv = (size_t)rd.integer();
            double fov = rd.number();
            sc->camera = std::make_unique<Camera>(h, vv, fov);
            sc->camera->transform = rd.lookup(rd.xfs, rd.word());
        } else if (cmd == "light") {
            PointLight l;
            double x = rd.number(), y = rd.number(), z = rd.number();
            l.position = pt(x, y, z);
            l.intensity.red = rd.number(), l.intensity.green = rd.number(), l.intensity.blue = rd.number();
            sc->world.lights.push_back(l);
        } else if (cmd == "objects") {
            int n = rd.integer();
            for (int k = 0; k < n; k++) sc->world.objects.push_back(rd.node());
        } else if (cmd == "end")
            break;
        else
            throw Error("sg: unknown record " + cmd);
    }
    if (!sc->camera) throw Error("sg: no camera record");
    return sc;
}

}  // namespace rt
