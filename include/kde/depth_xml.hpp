// kde/depth_xml.hpp — minimal reader for the reference's depth file: an OpenCV FileStorage XML with
// cv::Mat_<float> nodes "averaged_depth" and "depth" (written at main.cpp:112-114, read at main.cpp:146-149).
// Header-only, no OpenCV.  Throws std::runtime_error on malformed input.
#ifndef KDE_DEPTH_XML_HPP
#define KDE_DEPTH_XML_HPP

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace kde {

struct DepthMatrix {
    int rows = 0, cols = 0;
    std::vector<float> data;
};

inline std::string xml_between(const std::string& s, const std::string& open, const std::string& close, size_t from = 0)
{
    const size_t a = s.find(open, from);
    if (a == std::string::npos) throw std::runtime_error("depth xml: missing " + open);
    const size_t b = s.find(close, a + open.size());
    if (b == std::string::npos) throw std::runtime_error("depth xml: missing " + close);
    return s.substr(a + open.size(), b - a - open.size());
}

inline DepthMatrix read_opencv_matrix(const std::string& text, const std::string& name)
{
    const std::string open = "<" + name + " type_id=\"opencv-matrix\">";
    const std::string body = xml_between(text, open, "</" + name + ">");
    DepthMatrix m;
    m.rows = std::atoi(xml_between(body, "<rows>", "</rows>").c_str());
    m.cols = std::atoi(xml_between(body, "<cols>", "</cols>").c_str());
    std::string dt = xml_between(body, "<dt>", "</dt>");
    if (dt.find('f') == std::string::npos && dt.find('d') == std::string::npos)
        throw std::runtime_error("depth xml: <" + name + "> is not a float matrix");
    std::istringstream in(xml_between(body, "<data>", "</data>"));
    std::string tok;
    m.data.reserve((size_t)m.rows * m.cols);
    while (in >> tok) {
        float v;
        if (tok == ".Inf" || tok == "+.Inf") v = std::numeric_limits<float>::infinity();
        else if (tok == "-.Inf") v = -std::numeric_limits<float>::infinity();
        else if (tok == ".Nan" || tok == "-.Nan" || tok == ".NaN") v = std::numeric_limits<float>::quiet_NaN();
        else v = std::strtof(tok.c_str(), nullptr);
        m.data.push_back(v);
    }
    if (m.rows <= 0 || m.cols <= 0 || m.data.size() != (size_t)m.rows * m.cols)
        throw std::runtime_error("depth xml: <" + name + "> has the wrong number of values");
    return m;
}

// depth + averaged_depth as main.cpp:146-149 reads them
inline void read_depth_xml(const std::string& path, DepthMatrix& depth, DepthMatrix& averaged_depth)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("depth xml: cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    if (text.find("<opencv_storage>") == std::string::npos) throw std::runtime_error("depth xml: not an OpenCV FileStorage file");
    averaged_depth = read_opencv_matrix(text, "averaged_depth");
    depth = read_opencv_matrix(text, "depth");
}

}  // namespace kde
#endif
