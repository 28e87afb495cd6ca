// raisim/Yaml.hpp — the small part of the Yaml::Node interface raisimGymTorch environments use for their cfg.yaml
// [RECALL raisimGymTorch/env/Yaml.hpp (mini-yaml), absent from /root/reference]: nested maps of scalars,
//   cfg["reward"]["forwardVel"]["coeff"].As<double>(),  cfg["num_envs"].template As<int>(),  node.IsNone(), iteration.
// Parser: block-style maps by indentation, `key: value` scalars, `#` comments, quoted strings.  No sequences, anchors
// or flow style (rsg_anymal's cfg.yaml needs none of them).
#pragma once

#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace Yaml {

class Node {
 public:
  Node() = default;
  bool IsNone() const { return !isMap_ && !isScalar_; }
  bool IsMap() const { return isMap_; }
  bool IsScalar() const { return isScalar_; }
  size_t Size() const { return keys_.size(); }
  /// missing keys yield a None node (upstream behaviour), so that As<T>(default) can be used
  const Node& operator[](const std::string& key) const {
    static const Node none;
    auto it = children_.find(key);
    return it == children_.end() ? none : *it->second;
  }
  Node& operator[](const std::string& key) {
    auto it = children_.find(key);
    if (it != children_.end()) return *it->second;
    isMap_ = true;
    keys_.push_back(key);
    return *(children_[key] = std::make_shared<Node>());
  }
  template <typename T>
  T As() const {
    if (!isScalar_) throw std::runtime_error("Yaml::Node::As: not a scalar");
    return convert<T>(scalar_);
  }
  template <typename T>
  T As(const T& defaultValue) const { return isScalar_ ? convert<T>(scalar_) : defaultValue; }
  Node& operator=(const std::string& v) { isScalar_ = true; scalar_ = v; return *this; }
  const std::vector<std::string>& Keys() const { return keys_; }   ///< insertion order

 private:
  template <typename T>
  static T convert(const std::string& s) {
    if constexpr (std::is_same<T, std::string>::value) return s;
    else if constexpr (std::is_same<T, bool>::value) return s == "true" || s == "True" || s == "yes" || s == "1";
    else {
      std::istringstream is(s);
      T v{};
      is >> v;
      if (is.fail()) throw std::runtime_error("Yaml::Node::As: cannot convert '" + s + "'");
      return v;
    }
  }
  bool isMap_ = false, isScalar_ = false;
  std::string scalar_;
  std::vector<std::string> keys_;
  std::map<std::string, std::shared_ptr<Node>> children_;
  friend void Parse(Node&, const std::string&);
};

/// parse YAML text (block maps + scalars) into root
inline void Parse(Node& root, const std::string& text) {
  root = Node();
  std::vector<std::pair<int, Node*>> stack;   // (indent of the map's keys, map node)
  stack.emplace_back(-1, &root);
  std::istringstream in(text);
  std::string line;
  int lineNo = 0;
  while (std::getline(in, line)) {
    ++lineNo;
    // strip comments outside quotes
    bool q = false; char qc = 0;
    for (size_t i = 0; i < line.size(); ++i) {
      if (!q && (line[i] == '"' || line[i] == '\'')) { q = true; qc = line[i]; }
      else if (q && line[i] == qc) q = false;
      else if (!q && line[i] == '#') { line.erase(i); break; }
    }
    size_t first = line.find_first_not_of(" \t\r");
    if (first == std::string::npos) continue;
    const int indent = (int)first;
    const size_t colon = line.find(':', first);
    if (colon == std::string::npos) throw std::runtime_error("Yaml::Parse: line " + std::to_string(lineNo) + ": expected 'key: value'");
    std::string key = line.substr(first, colon - first);
    while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
    std::string val = line.substr(colon + 1);
    const size_t vb = val.find_first_not_of(" \t\r");
    val = vb == std::string::npos ? "" : val.substr(vb);
    while (!val.empty() && (val.back() == ' ' || val.back() == '\t' || val.back() == '\r')) val.pop_back();
    if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front()) val = val.substr(1, val.size() - 2);
    while (stack.size() > 1 && indent <= stack.back().first) stack.pop_back();
    Node& parent = *stack.back().second;
    Node& child = parent[key];
    if (val.empty()) stack.emplace_back(indent, &child);   // a nested map follows (or an empty value)
    else child = val;
  }
}

}  // namespace Yaml
