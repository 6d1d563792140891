// scene_yaml.hpp — Scene(filename): the scene-file loader of the reference (ref src/scene.cpp:32-189) without OpenCV.
//
// The reference reads its scene files with cv::FileStorage (YAML 1.0 dialect). This header parses the subset of YAML
// those files use — block mappings by indentation, block sequences of mappings ("- key: value"), flow sequences
// "[a, b, c]", plain / quoted scalars, "#" comments, the "%YAML:1.0" directive and "---" — and then builds the scene
// with the reference's schema and conversions:
//   colors      name: [r, g, b] | grey scalar                                   (scene.cpp:32-51)
//   materials   name: {color, ka, kd, ks, kr, kt, specRefExp, refractiveIdx, reflective, refractive}; an unknown key
//               throws "Not supported type"; booleans are true only for the text "true"   (scene.cpp:59-101)
//   lights      - pos, color, factor (default 1.0): Light{pos, color * factor}   (scene.cpp:112-127)
//   primitives  - type: sphere {radius, center} | mesh {translation, rotation, scale, filename} | plane {normal,
//               distance}, each with `material`; a mesh is load_ply(filename, material, true), *= Rz.Ry.Rx(rotation),
//               scale(double), += translation; any other type throws "Type is not supported"   (scene.cpp:136-178)
// Numbers are read as the reference's reader hands them over: integers and reals parsed in double and narrowed to
// Float where the reference assigns to a Float (materials, radius, light factor), kept double where it casts to
// double (mesh scale, plane distance).
// Deviations, all on inputs the reference has undefined behaviour for: a material / colour name that does not exist
// throws std::runtime_error (the reference dereferences map::end()).
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "raytracer.hpp"

namespace rt_yaml {

struct Node {
    enum Kind { NONE, SCALAR, SEQ, MAP } kind = NONE;
    std::string scalar;                                        // SCALAR
    std::vector<Node> seq;                                     // SEQ
    std::vector<std::pair<std::string, Node>> map;             // MAP, file order
    const Node* find(const std::string& key) const {
        for (const auto& kv : map) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Node& at(const std::string& key) const {
        const Node* n = find(key);
        if (!n) throw std::runtime_error("scene file: key '" + key + "' missing");
        return *n;
    }
    double number() const {
        if (kind != SCALAR) throw std::runtime_error("scene file: number expected");
        char* end = nullptr;
        const double v = std::strtod(scalar.c_str(), &end);
        if (end == scalar.c_str()) throw std::runtime_error("scene file: '" + scalar + "' is not a number");
        return v;
    }
};

struct Line { int indent; std::string text; int no; };

inline std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}

// Strips a trailing comment ('#' outside quotes, at line start or after white space).
inline std::string strip_comment(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; continue; }
        if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}

inline std::string unquote(const std::string& s) {
    if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) return s.substr(1, s.size() - 2);
    return s;
}

// "value" part of "key: value" or of a sequence item: flow sequence or scalar.
inline Node parse_inline(const std::string& text, int line_no) {
    Node n;
    const std::string t = trim(text);
    if (!t.empty() && t.front() == '[') {
        if (t.back() != ']') throw std::runtime_error("scene file line " + std::to_string(line_no) + ": flow sequence must close on the same line");
        n.kind = Node::SEQ;
        std::string item;
        std::stringstream ss(t.substr(1, t.size() - 2));
        while (std::getline(ss, item, ',')) {
            const std::string v = trim(item);
            if (v.empty()) continue;
            Node e; e.kind = Node::SCALAR; e.scalar = unquote(v);
            n.seq.push_back(e);
        }
        return n;
    }
    n.kind = Node::SCALAR;
    n.scalar = unquote(t);
    return n;
}

// Splits "key: value" at the first ':' that is followed by a space or ends the line (outside quotes / brackets).
inline bool split_key(const std::string& t, std::string& key, std::string& value) {
    char q = 0;
    int depth = 0;
    for (size_t i = 0; i < t.size(); ++i) {
        const char c = t[i];
        if (q) { if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; continue; }
        if (c == '[') ++depth;
        if (c == ']') --depth;
        if (c == ':' && depth == 0 && (i + 1 == t.size() || t[i + 1] == ' ' || t[i + 1] == '\t')) {
            key = unquote(trim(t.substr(0, i)));
            value = trim(t.substr(i + 1));
            return true;
        }
    }
    return false;
}

class Parser {
public:
    explicit Parser(std::istream& in) {
        std::string raw;
        int no = 0;
        while (std::getline(in, raw)) {
            ++no;
            const std::string s = strip_comment(raw);
            const std::string t = trim(s);
            if (t.empty() || t[0] == '%' || t == "---" || t == "...") continue;
            int indent = 0;
            while (indent < (int)s.size() && s[indent] == ' ') ++indent;
            if (indent < (int)s.size() && s[indent] == '\t') throw std::runtime_error("scene file line " + std::to_string(no) + ": tab indentation");
            lines_.push_back({indent, t, no});
        }
    }
    Node document() {
        pos_ = 0;
        if (lines_.empty()) return Node{};
        Node n = block(lines_[0].indent);
        if (pos_ != lines_.size()) throw std::runtime_error("scene file line " + std::to_string(lines_[pos_].no) + ": unexpected indentation");
        return n;
    }

private:
    // Block (mapping or sequence) whose entries start at column `indent`.
    Node block(int indent) {
        Node n;
        const bool is_seq = lines_[pos_].text[0] == '-' && (lines_[pos_].text.size() == 1 || lines_[pos_].text[1] == ' ');
        n.kind = is_seq ? Node::SEQ : Node::MAP;
        while (pos_ < lines_.size() && lines_[pos_].indent == indent) {
            const Line ln = lines_[pos_];
            if (is_seq) {
                if (!(ln.text[0] == '-' && (ln.text.size() == 1 || ln.text[1] == ' '))) break;
                // "- key: value": the item is a mapping whose entries sit at the column of `key`
                size_t off = 1;
                while (off < ln.text.size() && ln.text[off] == ' ') ++off;
                const std::string rest = ln.text.substr(off);
                std::string key, value;
                if (rest.empty()) {                       // "-" alone: nested block on the following lines
                    ++pos_;
                    if (pos_ >= lines_.size() || lines_[pos_].indent <= indent) throw std::runtime_error("scene file line " + std::to_string(ln.no) + ": empty sequence item");
                    n.seq.push_back(block(lines_[pos_].indent));
                } else if (split_key(rest, key, value)) {
                    lines_[pos_].indent = indent + (int)off;   // re-read this line as the first entry of the item's mapping
                    lines_[pos_].text = rest;
                    n.seq.push_back(block(indent + (int)off));
                } else {
                    n.seq.push_back(parse_inline(rest, ln.no));
                    ++pos_;
                }
            } else {
                std::string key, value;
                if (!split_key(ln.text, key, value)) throw std::runtime_error("scene file line " + std::to_string(ln.no) + ": 'key: value' expected");
                ++pos_;
                if (!value.empty()) {
                    n.map.emplace_back(key, parse_inline(value, ln.no));
                } else if (pos_ < lines_.size() && lines_[pos_].indent > indent) {
                    n.map.emplace_back(key, block(lines_[pos_].indent));
                } else if (pos_ < lines_.size() && lines_[pos_].indent == indent && lines_[pos_].text[0] == '-' &&
                           (lines_[pos_].text.size() == 1 || lines_[pos_].text[1] == ' ')) {
                    n.map.emplace_back(key, block(indent));      // a sequence may sit at its key's own column
                } else {
                    n.map.emplace_back(key, Node{});
                }
            }
        }
        if (pos_ < lines_.size() && lines_[pos_].indent > indent)
            throw std::runtime_error("scene file line " + std::to_string(lines_[pos_].no) + ": unexpected indentation");
        return n;
    }
    std::vector<Line> lines_;
    size_t pos_ = 0;
};

inline Node parse_file(const std::string& filename) {
    std::ifstream in(filename);
    if (!in.good()) throw std::runtime_error("Scene file " + filename + " could not be opened.");
    Parser p(in);
    return p.document();
}

inline Vec3f vec3(const Node& n) {              // read(node, vec): elements 0..2 (scene.cpp:25-30)
    if (n.kind != Node::SEQ || n.seq.size() < 3) throw std::runtime_error("scene file: sequence of 3 numbers expected");
    return Vec3f{Float(n.seq[0].number()), Float(n.seq[1].number()), Float(n.seq[2].number())};
}

}  // namespace rt_yaml

inline Scene::Scene(const std::string& filename) {
    using rt_yaml::Node;
    const Node doc = rt_yaml::parse_file(filename);
    const Node none;

    std::map<std::string, Color> colors;                                          // scene.cpp:32-51
    if (const Node* cs = doc.find("colors"))
        for (const auto& kv : cs->map)
            colors[kv.first] = kv.second.kind == Node::SEQ ? Color{rt_yaml::vec3(kv.second)} : Color{Float(kv.second.number())};
    auto color_of = [&](const Node& n) -> const Color& {
        const auto it = colors.find(n.scalar);
        if (n.kind != Node::SCALAR || it == colors.end()) throw std::runtime_error("scene file: unknown color '" + n.scalar + "'");
        return it->second;
    };

    std::map<std::string, Material> materials;                                    // scene.cpp:59-101
    if (const Node* ms = doc.find("materials"))
        for (const auto& kv : ms->map) {
            Material m;
            for (const auto& e : kv.second.map) {
                const std::string& k = e.first;
                if (k == "color") m.color = color_of(e.second);
                else if (k == "ka") m.ka = Float(e.second.number());
                else if (k == "kd") m.kd = Float(e.second.number());
                else if (k == "ks") m.ks = Float(e.second.number());
                else if (k == "kr") m.kr = Float(e.second.number());
                else if (k == "kt") m.kt = Float(e.second.number());
                else if (k == "specRefExp") m.specRefExp = Float(e.second.number());
                else if (k == "refractiveIdx") m.refractiveIdx = Float(e.second.number());
                else if (k == "reflective") m.reflective = e.second.scalar == "true";
                else if (k == "refractive") m.refractive = e.second.scalar == "true";
                else throw std::runtime_error{"Not supported type"};
            }
            materials[kv.first] = m;
        }

    if (const Node* ls = doc.find("lights"))                                      // scene.cpp:112-127
        for (const Node& l : ls->seq) {
            const Vec3f pos = rt_yaml::vec3(l.at("pos"));
            float factor = 1.0f;
            if (const Node* f = l.find("factor")) factor = float(f->number());
            lights.push_back(Light{pos, color_of(l.at("color")) * factor});
        }

    if (const Node* ps = doc.find("primitives"))                                  // scene.cpp:136-178
        for (const Node& p : ps->seq) {
            const std::string type = p.at("type").scalar;
            const auto mit = materials.find(p.at("material").scalar);
            if (mit == materials.end()) throw std::runtime_error("scene file: unknown material '" + p.at("material").scalar + "'");
            const Material& material = mit->second;
            if (type == "sphere") {
                const float radius = static_cast<float>(p.at("radius").number());
                objects.emplace_back(new Sphere{rt_yaml::vec3(p.at("center")), radius, material});
            } else if (type == "mesh") {
                const Vec3f translation = rt_yaml::vec3(p.at("translation")), rotation = rt_yaml::vec3(p.at("rotation"));
                const double scale = p.at("scale").number();
                const std::string fname = p.at("filename").scalar;
                Model* model = Model::load_ply(fname, material, true);
                if (model == nullptr) throw std::runtime_error{"Couldn't load model."};
                *model *= Mat3d::rotation(rotation);
                model->scale(scale);
                *model += translation;
                objects.emplace_back(model);
            } else if (type == "plane") {
                const double d = p.at("distance").number();
                objects.emplace_back(new Plane{rt_yaml::vec3(p.at("normal")), Float(d), material});
            } else {
                throw std::runtime_error{"Type is not supported"};
            }
        }
}

