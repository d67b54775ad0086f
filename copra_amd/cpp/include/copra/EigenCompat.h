// EigenCompat.h -- the dense types copra's public API is written in.
// With Eigen3 installed this is just <Eigen/Core>.  The build image has no Eigen, so a minimal column-major
// MatrixXd / VectorXd with the handful of members copra's API and tests use is provided under the same names; it is a
// container (storage + indexing + comma initialiser), not a linear-algebra library -- all arithmetic of the MPC path
// runs on the GPU behind include/copra_hip.h.
#pragma once
#if defined(__has_include)
#if __has_include(<Eigen/src/Core/Matrix.h>) // (the real Eigen3, not copra/eigen_shim/Eigen/Core, which leads back here)
#include <Eigen/Core>
#define COPRA_HAVE_EIGEN 1
#endif
#endif

#ifndef COPRA_HAVE_EIGEN
#include <cstddef>
#include <initializer_list>
#include <limits>
#include <vector>

namespace Eigen {
using Index = std::ptrdiff_t;

class MatrixXd;
// "m << 1, 2, 3, 4;" fills row by row like Eigen's CommaInitializer
template <class M>
class CommaInit {
public:
    CommaInit(M& m, double v)
        : m_(m)
        , k_(0)
    {
        put(v);
    }
    CommaInit& operator,(double v)
    {
        put(v);
        return *this;
    }

private:
    void put(double v)
    {
        const Index c = m_.cols() > 0 ? m_.cols() : 1;
        m_(k_ / c, k_ % c) = v;
        ++k_;
    }
    M& m_;
    Index k_;
};

class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(Index r, Index c)
        : r_(r)
        , c_(c)
        , v_((size_t)(r * c), 0.0)
    {
    }
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    Index size() const { return r_ * c_; }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    double& operator()(Index i, Index j) { return v_[(size_t)(j * r_ + i)]; }
    double operator()(Index i, Index j) const { return v_[(size_t)(j * r_ + i)]; }
    void resize(Index r, Index c)
    {
        r_ = r;
        c_ = c;
        v_.assign((size_t)(r * c), 0.0);
    }
    void setZero() { v_.assign(v_.size(), 0.0); }
    void setConstant(double x) { v_.assign(v_.size(), x); }
    void setIdentity()
    {
        setZero();
        for (Index i = 0; i < (r_ < c_ ? r_ : c_); ++i) (*this)(i, i) = 1.0;
    }
    CommaInit<MatrixXd> operator<<(double v) { return CommaInit<MatrixXd>(*this, v); }
    static MatrixXd Zero(Index r, Index c) { return MatrixXd(r, c); }
    static MatrixXd Ones(Index r, Index c)
    {
        MatrixXd m(r, c);
        m.setConstant(1.0);
        return m;
    }
    static MatrixXd Identity(Index r, Index c)
    {
        MatrixXd m(r, c);
        m.setIdentity();
        return m;
    }
    MatrixXd operator*(double s) const
    {
        MatrixXd m(*this);
        for (auto& x : m.v_) x *= s;
        return m;
    }

private:
    Index r_ = 0, c_ = 0;
    std::vector<double> v_;
};

class VectorXd {
public:
    VectorXd() = default;
    explicit VectorXd(Index n)
        : v_((size_t)n, 0.0)
    {
    }
    Index rows() const { return (Index)v_.size(); }
    Index cols() const { return 1; }
    Index size() const { return (Index)v_.size(); }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    double& operator()(Index i) { return v_[(size_t)i]; }
    double operator()(Index i) const { return v_[(size_t)i]; }
    double& operator()(Index i, Index) { return v_[(size_t)i]; } // for the comma initialiser
    double& operator[](Index i) { return v_[(size_t)i]; }
    double operator[](Index i) const { return v_[(size_t)i]; }
    void resize(Index n) { v_.assign((size_t)n, 0.0); }
    void setZero() { v_.assign(v_.size(), 0.0); }
    void setConstant(double x) { v_.assign(v_.size(), x); }
    void setConstant(Index n, double x) { v_.assign((size_t)n, x); }
    CommaInit<VectorXd> operator<<(double v) { return CommaInit<VectorXd>(*this, v); }
    static VectorXd Zero(Index n) { return VectorXd(n); }
    static VectorXd Ones(Index n)
    {
        VectorXd v(n);
        v.setConstant(1.0);
        return v;
    }
    static VectorXd Constant(Index n, double x)
    {
        VectorXd v(n);
        v.setConstant(x);
        return v;
    }
    VectorXd head(Index n) const
    {
        VectorXd o(n);
        for (Index i = 0; i < n; ++i) o(i) = v_[(size_t)i];
        return o;
    }
    VectorXd tail(Index n) const
    {
        VectorXd o(n);
        for (Index i = 0; i < n; ++i) o(i) = v_[v_.size() - (size_t)n + (size_t)i];
        return o;
    }
    VectorXd segment(Index at, Index n) const
    {
        VectorXd o(n);
        for (Index i = 0; i < n; ++i) o(i) = v_[(size_t)(at + i)];
        return o;
    }
    double maxCoeff() const
    {
        double m = -std::numeric_limits<double>::infinity();
        for (double x : v_) m = x > m ? x : m;
        return m;
    }
    double minCoeff() const
    {
        double m = std::numeric_limits<double>::infinity();
        for (double x : v_) m = x < m ? x : m;
        return m;
    }
    VectorXd operator*(double s) const
    {
        VectorXd o(*this);
        for (auto& x : o.v_) x *= s;
        return o;
    }

private:
    std::vector<double> v_;
};
inline VectorXd operator*(double s, const VectorXd& v) { return v * s; }
inline MatrixXd operator*(double s, const MatrixXd& m) { return m * s; }
} // namespace Eigen
#endif
