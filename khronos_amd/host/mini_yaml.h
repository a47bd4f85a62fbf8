// mini_yaml.h — the YAML subset the Khronos mapper configs use (khronos_ros/config/mapper/*.yaml):
// block mappings by indentation, scalars (numbers / bools / quoted or plain strings), comments, anchors
// (&name value) and aliases (*name), empty flow sequences ("[]"), block sequences ("- type: X" items: the entries of a
// list are children with the key "-", see YamlNode::items).  Stands in for config_utilities'
// YAML front end, which is not available offline (SURVEY.md §5 "Config / flag system").
#pragma once
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace khronos_amd {

struct YamlNode {
  bool is_map = false;
  std::string scalar;
  std::vector<std::pair<std::string, YamlNode>> children;  // insertion order

  const YamlNode* find(const std::string& key) const {
    for (const auto& kv : children)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  bool has(const std::string& key) const { return find(key) != nullptr; }
  // the entries of a block sequence ("- ..." lines), in order
  std::vector<const YamlNode*> items() const {
    std::vector<const YamlNode*> out;
    for (const auto& kv : children)
      if (kv.first == "-") out.push_back(&kv.second);
    return out;
  }
  const YamlNode& at(const std::string& key) const {
    const YamlNode* n = find(key);
    if (!n) throw std::runtime_error("yaml: missing key '" + key + "'");
    return *n;
  }
  template <typename T>
  void read(const std::string& key, T& out) const;
};

namespace detail {
inline std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::string stripComment(const std::string& s) {
  bool in_s = false, in_d = false;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '\'' && !in_d) in_s = !in_s;
    if (s[i] == '"' && !in_s) in_d = !in_d;
    if (s[i] == '#' && !in_s && !in_d && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  }
  return s;
}
inline std::string unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
    return s.substr(1, s.size() - 2);
  return s;
}
}  // namespace detail

namespace detail {
// position of the ':' that separates a mapping key from its value: outside quotes and followed by a space or the end of the
// line ("12:30", "http://host" and "'a: b'" are scalars) -- npos if there is none
inline size_t keyColon(const std::string& s) {
  char quote = 0;
  for (size_t i = 0; i < s.size(); ++i) {
    const char ch = s[i];
    if (quote) {
      if (ch == quote) quote = 0;
    } else if (ch == '"' || ch == '\'') {
      quote = ch;
    } else if (ch == ':' && (i + 1 == s.size() || s[i + 1] == ' ' || s[i + 1] == '\t')) {
      return i;
    }
  }
  return std::string::npos;
}
}  // namespace detail

inline YamlNode parseYaml(const std::string& text) {
  struct Line {
    int indent;
    std::string key, value;
  };
  std::vector<Line> lines;
  std::istringstream in(text);
  std::string raw;
  while (std::getline(in, raw)) {
    std::string s = detail::stripComment(raw);
    if (detail::trim(s).empty()) continue;
    const int indent = static_cast<int>(s.find_first_not_of(' '));
    s = detail::trim(s);
    if (s == "---") continue;
    // block sequence entry: "- key: value" opens an item (a mapping whose first key sits two columns further in),
    // "- scalar" is a scalar item
    int extra = 0;
    if (s == "-" || s.rfind("- ", 0) == 0) {
      const std::string rest = detail::trim(s.substr(1));
      if (rest.empty() || detail::keyColon(rest) != std::string::npos) {
        lines.push_back({indent, "-", ""});
        if (rest.empty()) continue;
        s = rest;
        extra = 2;
      } else {
        lines.push_back({indent, "-", rest});
        continue;
      }
    }
    const size_t colon = detail::keyColon(s);
    if (colon == std::string::npos) throw std::runtime_error("yaml: unsupported line '" + s + "'");
    lines.push_back({indent + extra, detail::trim(s.substr(0, colon)), detail::trim(s.substr(colon + 1))});
  }
  std::map<std::string, YamlNode> anchors;
  size_t pos = 0;
  // recursive descent over indentation
  struct Rec {
    std::vector<Line>& L;
    std::map<std::string, YamlNode>& A;
    size_t& pos;
    YamlNode block(int indent) {
      YamlNode node;
      node.is_map = true;
      while (pos < L.size() && L[pos].indent == indent) {
        const Line ln = L[pos++];
        std::string v = ln.value, anchor;
        if (!v.empty() && v[0] == '&') {
          const size_t sp = v.find(' ');
          anchor = v.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
          v = sp == std::string::npos ? std::string() : detail::trim(v.substr(sp + 1));
        }
        YamlNode child;
        if (v.empty()) {
          if (pos < L.size() && L[pos].indent > indent) child = block(L[pos].indent);
          else child.is_map = true;  // empty mapping
        } else if (v[0] == '*') {
          auto it = A.find(v.substr(1));
          if (it == A.end()) throw std::runtime_error("yaml: unknown alias '" + v + "'");
          child = it->second;
        } else {
          child.scalar = detail::unquote(v);
        }
        if (!anchor.empty()) A[anchor] = child;
        node.children.emplace_back(detail::unquote(ln.key), child);
      }
      if (pos < L.size() && L[pos].indent > indent) throw std::runtime_error("yaml: bad indentation near '" + L[pos].key + "'");
      return node;
    }
  } rec{lines, anchors, pos};
  YamlNode root = lines.empty() ? YamlNode() : rec.block(lines[0].indent);
  root.is_map = true;
  return root;
}

template <>
inline void YamlNode::read<std::string>(const std::string& key, std::string& out) const {
  if (const YamlNode* n = find(key)) out = n->scalar;
}
template <>
inline void YamlNode::read<float>(const std::string& key, float& out) const {
  if (const YamlNode* n = find(key)) out = std::strtof(n->scalar.c_str(), nullptr);
}
template <>
inline void YamlNode::read<double>(const std::string& key, double& out) const {
  if (const YamlNode* n = find(key)) out = std::strtod(n->scalar.c_str(), nullptr);
}
template <>
inline void YamlNode::read<int>(const std::string& key, int& out) const {
  if (const YamlNode* n = find(key)) out = static_cast<int>(std::strtol(n->scalar.c_str(), nullptr, 10));
}
template <>
inline void YamlNode::read<size_t>(const std::string& key, size_t& out) const {
  if (const YamlNode* n = find(key)) out = static_cast<size_t>(std::strtoull(n->scalar.c_str(), nullptr, 10));
}
template <>
inline void YamlNode::read<bool>(const std::string& key, bool& out) const {
  if (const YamlNode* n = find(key)) out = (n->scalar == "true" || n->scalar == "True" || n->scalar == "1");
}

}  // namespace khronos_amd
