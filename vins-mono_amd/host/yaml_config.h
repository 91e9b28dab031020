// yaml_config.h — dependency-free reader for the YAML subset of the VINS-Mono configuration files
// (config/euroc/euroc_config.yaml and friends), replacing the cv::FileStorage the reference opens in
// feature_tracker/src/parameters.cpp:37-74, vins_estimator/src/parameters.cpp:42-137 and camodocal's
// PinholeCamera::Parameters::readFromYamlFile (camera_model/src/camera_models/PinholeCamera.cc).
//
// Subset (everything those files use):
//   %YAML:1.0 header, `#` comments (also trailing), blank lines;
//   top-level scalars     key: value            numbers, bare words, "quoted strings";
//   one level of mapping  key:\n   child: value   -> addressed as "key.child" (distortion_parameters.k1 ...);
//   OpenCV matrices       key: !!opencv-matrix\n   rows: R\n   cols: C\n   dt: d\n   data: [ ... ]   (data may span lines).
// Header-only; used by the host shims (readParameters / readIntrinsicParameter) and — through the cv::FileStorage stand-in of
// oracle/ref_stubs — by the reference's own parameters.cpp in the parity test.
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

class VinsYaml {
  public:
    struct Matrix {
        int rows = 0, cols = 0;
        std::string dt;
        std::vector<double> data;      // row-major
    };

    bool load(const std::string& path) {
        std::ifstream f(path.c_str());
        if (!f.is_open()) return false;
        std::stringstream ss;
        ss << f.rdbuf();
        return parse(ss.str());
    }

    bool parse(const std::string& text) {
        scalars_.clear(); matrices_.clear(); opened_ = false;
        std::vector<std::string> lines;
        {
            std::stringstream ss(text);
            std::string l;
            while (std::getline(ss, l)) lines.push_back(strip_comment(l));
        }
        std::string parent;            // key of the mapping / matrix whose indented children are being read ("" at top level)
        int parent_indent = -1;
        Matrix* mat = nullptr;         // non-null while the parent is an !!opencv-matrix
        for (size_t i = 0; i < lines.size(); ++i) {
            const std::string& raw = lines[i];
            const std::string t = trim(raw);
            if (t.empty() || t[0] == '%' || t == "---") continue;
            const int indent = (int)raw.find_first_not_of(" \t");
            const size_t colon = find_colon(t);
            if (colon == std::string::npos) return false;                 // not `key: ...`
            const std::string key = trim(t.substr(0, colon));
            const std::string val = trim(t.substr(colon + 1));
            if (!parent.empty() && indent > parent_indent) {              // a child of the current mapping
                if (mat) {
                    if (key == "rows") mat->rows = atoi(val.c_str());
                    else if (key == "cols") mat->cols = atoi(val.c_str());
                    else if (key == "dt") mat->dt = unquote(val);
                    else if (key == "data") {
                        std::string list = val;
                        while (list.find(']') == std::string::npos && i + 1 < lines.size()) list += " " + trim(lines[++i]);
                        if (!parse_list(list, mat->data)) return false;
                    }
                } else {
                    scalars_[parent + "." + key] = unquote(val);
                }
                continue;
            }
            parent.clear(); parent_indent = -1; mat = nullptr;            // back at the top level
            if (val.empty()) {                                            // a mapping follows
                parent = key; parent_indent = indent;
            } else if (val.compare(0, 15, "!!opencv-matrix") == 0) {
                matrices_[key] = Matrix();
                parent = key; parent_indent = indent; mat = &matrices_[key];
            } else {
                scalars_[key] = unquote(val);
            }
        }
        for (auto& kv : matrices_)
            if (kv.second.rows * kv.second.cols != (int)kv.second.data.size()) return false;
        opened_ = true;
        return true;
    }

    bool opened() const { return opened_; }
    bool has(const std::string& key) const { return scalars_.count(key) || matrices_.count(key); }
    bool is_matrix(const std::string& key) const { return matrices_.count(key) != 0; }
    std::string str(const std::string& key, const std::string& dflt = "") const {
        auto it = scalars_.find(key);
        return it == scalars_.end() ? dflt : it->second;
    }
    // cv::FileNode semantics: a missing or non-numeric node converts to 0
    double number(const std::string& key, double dflt = 0.0) const {
        auto it = scalars_.find(key);
        if (it == scalars_.end()) return dflt;
        char* end = nullptr;
        const double v = strtod(it->second.c_str(), &end);
        return end == it->second.c_str() ? dflt : v;
    }
    const Matrix* matrix(const std::string& key) const {
        auto it = matrices_.find(key);
        return it == matrices_.end() ? nullptr : &it->second;
    }

  private:
    std::map<std::string, std::string> scalars_;
    std::map<std::string, Matrix> matrices_;
    bool opened_ = false;

    static std::string trim(const std::string& s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::string strip_comment(const std::string& s) {
        bool q = false;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == '"') q = !q;
            if (s[i] == '#' && !q) return s.substr(0, i);
        }
        return s;
    }
    static size_t find_colon(const std::string& t) {
        bool q = false;
        for (size_t i = 0; i < t.size(); ++i) {
            if (t[i] == '"') q = !q;
            if (t[i] == ':' && !q && (i + 1 == t.size() || isspace((unsigned char)t[i + 1]))) return i;
        }
        return std::string::npos;
    }
    static std::string unquote(const std::string& v) {
        if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) return v.substr(1, v.size() - 2);
        return v;
    }
    static bool parse_list(const std::string& list, std::vector<double>& out) {
        const size_t a = list.find('['), b = list.find(']');
        if (a == std::string::npos || b == std::string::npos || b < a) return false;
        std::string body = list.substr(a + 1, b - a - 1);
        for (auto& ch : body) if (ch == ',') ch = ' ';
        std::stringstream ss(body);
        std::string tok;
        out.clear();
        while (ss >> tok) {
            char* end = nullptr;
            const double v = strtod(tok.c_str(), &end);
            if (end == tok.c_str()) return false;
            out.push_back(v);
        }
        return true;
    }
};
