// Host only.  == MomaTrajOpt::init (planner/include/planner/moma_traj_opt.h:845-941): the optimiser's parameters from the
// reference's parameter file (src/planner/params/optimizer.yaml, `planner_node: moma_traj_opt: ...`), for a C or C++
// caller that has no ROS parameter server (topay_amd/api.py: params_from_yaml is the same mapping for Python callers).
// The file is the plain subset of YAML the reference's parameter files use: nested mappings by indentation, scalars,
// one-line lists in brackets, comments.
#pragma once
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/topay.h"

namespace topay_yaml {

// flat view: "first_stage/lbfgs/mem_size" -> "256", "energy_weights" -> "0.33, 1.0, ..." (brackets removed)
inline bool parse(const std::string& text, std::map<std::string, std::string>& out, std::vector<std::string>& order, std::string& err) {
  std::vector<std::pair<int, std::string>> stack;   // (indent, key)
  std::istringstream in(text);
  std::string line;
  int ln = 0;
  while (std::getline(in, line)) {
    ln++;
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < line.size(); i++) {   // strip the comment
      if (line[i] == '\'' && !in_d) in_s = !in_s;
      else if (line[i] == '"' && !in_s) in_d = !in_d;
      else if (line[i] == '#' && !in_s && !in_d && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) { line.erase(i); break; }
    }
    size_t e = line.find_last_not_of(" \t\r\n");
    if (e == std::string::npos) continue;
    line.erase(e + 1);
    size_t ind = line.find_first_not_of(' ');
    if (line[ind] == '\t') { err = "line " + std::to_string(ln) + ": tab indentation"; return false; }
    if (line.compare(ind, 3, "---") == 0) continue;
    size_t colon = line.find(':', ind);
    if (colon == std::string::npos) { err = "line " + std::to_string(ln) + ": expected `key: value`"; return false; }
    std::string key = line.substr(ind, colon - ind);
    size_t ke = key.find_last_not_of(" \t");
    key.erase(ke + 1);
    std::string val = colon + 1 < line.size() ? line.substr(colon + 1) : "";
    size_t vs = val.find_first_not_of(" \t");
    val = vs == std::string::npos ? "" : val.substr(vs);
    while (!stack.empty() && stack.back().first >= (int)ind) stack.pop_back();
    if (val.empty()) { stack.emplace_back((int)ind, key); continue; }   // a nested mapping follows
    if (val[0] == '{') {   // one-line mapping: {k: v, k: v}
      size_t close = val.find('}');
      if (close == std::string::npos) { err = "line " + std::to_string(ln) + ": mapping not closed on its line"; return false; }
      std::string body = val.substr(1, close - 1), path;
      for (auto& s : stack) path += s.second + "/";
      path += key + "/";
      std::istringstream items(body);
      std::string item;
      while (std::getline(items, item, ',')) {
        size_t c2 = item.find(':');
        if (c2 == std::string::npos) continue;
        std::string k2 = item.substr(0, c2), v2 = item.substr(c2 + 1);
        auto trim = [](std::string& t) { size_t a = t.find_first_not_of(" \t"), b = t.find_last_not_of(" \t"); t = a == std::string::npos ? "" : t.substr(a, b - a + 1); };
        trim(k2); trim(v2);
        out[path + k2] = v2;
        order.push_back(path + k2);
      }
      continue;
    }
    if (val[0] == '[') {
      size_t close = val.find(']');
      if (close == std::string::npos) { err = "line " + std::to_string(ln) + ": list not closed on its line"; return false; }
      val = val.substr(1, close - 1);
    }
    if (val.size() >= 2 && ((val.front() == '"' && val.back() == '"') || (val.front() == '\'' && val.back() == '\''))) val = val.substr(1, val.size() - 2);
    std::string path;
    for (auto& s : stack) path += s.second + "/";
    path += key;
    out[path] = val;
    order.push_back(path);
  }
  return true;
}

inline std::vector<double> numbers(const std::string& v) {
  std::vector<double> r;
  std::string t = v;
  for (char& c : t) if (c == ',') c = ' ';
  std::istringstream in(t);
  std::string tok;
  while (in >> tok) r.push_back(strtod(tok.c_str(), nullptr));
  return r;
}

// Applies the file to *p (keys that are absent keep p's value).  `ignored` collects, separated by newlines, the keys
// init() reads but this path has no use for and any key init() does not know -- the same list api.params_from_yaml returns.
inline topay_status apply(const std::string& text, topay_params_t* p, std::string& ignored, std::string& err) {
  std::map<std::string, std::string> kv;
  std::vector<std::string> order;
  if (!parse(text, kv, order, err)) return TOPAY_ERR_INVALID_ARG;
  auto num = [&](const std::string& v) { return strtod(v.c_str(), nullptr); };
  auto vec = [&](double* dst, const std::string& v, int n) {
    std::vector<double> x = numbers(v);
    for (int i = 0; i < n && i < (int)x.size(); i++) dst[i] = x[i];
  };
  auto ign = [&](const std::string& k) { ignored += (ignored.empty() ? "" : "\n") + k; };
  auto lbfgs = [&](topay_lbfgs_params_t& L, const std::string& k, const std::string& v, const std::string& full, bool allow_past) {
    if (k == "mem_size") L.mem_size = (int)num(v);
    else if (k == "past" && allow_past) L.past = (int)num(v);
    else if (k == "max_iterations") L.max_iterations = (int)num(v);
    else if (k == "g_epsilon") L.g_epsilon = num(v);
    else if (k == "min_step") L.min_step = num(v);
    else if (k == "delta") L.delta = num(v);
    else ign(full);
  };
  for (const std::string& path : order) {
    std::string k = path;
    for (const char* pre : {"planner_node/", "moma_traj_opt/"})
      if (k.compare(0, strlen(pre), pre) == 0) k = k.substr(strlen(pre));
    const std::string& v = kv[path];
    if (k == "int_K") p->int_K = (int)num(v);
    else if (k == "min_piece_num") p->min_piece_num = (int)num(v);
    else if (k == "relu_mu") p->relu_mu = num(v);
    else if (k == "sample_interval") p->sample_interval = num(v);
    else if (k == "energy_weights") vec(p->energy_weights, v, 9);
    else if (k.compare(0, 12, "first_stage/") == 0) {
      const std::string kk = k.substr(12);
      if (kk == "time_weight") p->s1_time_weight = num(v);
      else if (kk == "moment_weight") p->s1_moment_weight = num(v);
      else if (kk == "acc_weight") p->s1_acc_weight = num(v);
      else if (kk == "domega_weight") p->s1_domega_weight = num(v);
      else if (kk == "path_pos_weight") p->s1_path_pos_weight = num(v);
      else if (kk == "lbgfs_normal_past") { p->s1_normal_past = (int)num(v); p->s1_lbfgs.past = (int)num(v); }   // (sic) moma_traj_opt.h:873
      else if (kk == "lbgfs_shot_path_past") p->s1_shot_path_past = (int)num(v);
      else if (kk == "shot_path_horizon") p->s1_shot_path_horizon = num(v);
      else if (kk.compare(0, 6, "lbfgs/") == 0) lbfgs(p->s1_lbfgs, kk.substr(6), v, k, false);
      else ign(k);
    } else if (k.compare(0, 13, "second_stage/") == 0) {
      const std::string kk = k.substr(13);
      if (kk == "time_weight") p->s2_time_weight = num(v);
      else if (kk == "moment_weight") p->s2_moment_weight = num(v);
      else if (kk == "acc_weight") p->s2_acc_weight = num(v);
      else if (kk == "domega_weight") p->s2_domega_weight = num(v);
      else if (kk == "collision_weight") p->s2_collision_weight = num(v);
      else if (kk == "mani_colli_weight") p->s2_mani_colli_weight = num(v);
      else if (kk == "self_colli_weight") p->s2_self_colli_weight = num(v);
      else if (kk == "mani_pos_weight") p->s2_mani_pos_weight = num(v);
      else if (kk == "mani_vel_weight") p->s2_mani_vel_weight = num(v);
      else if (kk == "mani_acc_weight") p->s2_mani_acc_weight = num(v);
      else if (kk == "mean_time_weight") p->s2_mean_time_weight = num(v);
      else if (kk.compare(0, 6, "lbfgs/") == 0) lbfgs(p->s2_lbfgs, kk.substr(6), v, k, true);
      else if (kk == "alm_param/init_lambda") vec(p->alm_init_lambda, v, 2);   // the reference sizes these 9 and uses entries 0 and 1
      else if (kk == "alm_param/init_rho") vec(p->alm_init_rho, v, 2);
      else if (kk == "alm_param/rho_max") vec(p->alm_rho_max, v, 2);
      else if (kk == "alm_param/gamma") vec(p->alm_gamma, v, 2);
      else if (kk == "alm_param/tolerance") vec(&p->alm_tolerance, v, 1);
      else ign(k);
    } else {
      ign(k);
    }
  }
  return TOPAY_OK;
}

inline topay_status apply_source(const char* path_or_text, topay_params_t* p, std::string& ignored, std::string& err) {
  std::string text;
  std::ifstream f(path_or_text);
  if (f.good() && !strchr(path_or_text, '\n')) {
    std::stringstream ss;
    ss << f.rdbuf();
    text = ss.str();
  } else {
    text = path_or_text;
  }
  return apply(text, p, ignored, err);
}

}  // namespace topay_yaml
