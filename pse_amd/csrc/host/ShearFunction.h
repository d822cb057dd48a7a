// Host-side shear functions of the PSE integrator, HOOMD-free.  Same class names, constructor arguments and method
// names as the reference (PSEv1/ShearFunction.h:24-39, PSEv1/SpecificShearFunction.h:16-223,
// PSEv1/VariantShearFunction.{h,cc}) so the pybind11 module and the Python UI read the same.
// Deliberate differences: exact pi instead of 3.1415926536; log instead of logf in the chirp;
// (timestep - offset) is a signed difference (the reference wraps around in unsigned arithmetic before the offset).
#pragma once
#include <cmath>
#include <memory>

namespace pse_host {

class ShearFunction {
public:
    virtual ~ShearFunction() = default;
    virtual double getShearRate(unsigned int timestep) { return 0.0; }   // ShearFunction.h:31
    virtual double getStrain(unsigned int timestep) { return 0.0; }      // ShearFunction.h:35
    virtual unsigned int getOffset() { return 0; }                       // ShearFunction.h:38
};

namespace detail {
inline double since(unsigned int timestep, unsigned int offset) { return (double)((long long)timestep - (long long)offset); }
constexpr double kPi = 3.14159265358979323846;
}  // namespace detail

// gamma_dot(t) = A cos(2 pi f t), gamma(t) = A sin(2 pi f t)/(2 pi f)   (SpecificShearFunction.h:31-36)
class SinShearFunction : public ShearFunction {
public:
    SinShearFunction(double max_shear_rate, double frequency, unsigned int offset, double dt)
        : m_max_shear_rate(max_shear_rate), m_frequency(frequency), m_offset(offset), m_dt(dt) {}
    double getShearRate(unsigned int t) override {
        return m_max_shear_rate * std::cos(m_frequency * 2 * detail::kPi * (detail::since(t, m_offset) * m_dt));
    }
    double getStrain(unsigned int t) override {
        return m_max_shear_rate * std::sin(m_frequency * 2 * detail::kPi * (detail::since(t, m_offset) * m_dt)) / m_frequency / 2 / detail::kPi;
    }
    unsigned int getOffset() override { return m_offset; }
private:
    const double m_max_shear_rate, m_frequency;
    const unsigned int m_offset;
    const double m_dt;
};

// constant rate (SpecificShearFunction.h:62-67)
class SteadyShearFunction : public ShearFunction {
public:
    SteadyShearFunction(double shear_rate, unsigned int offset, double dt) : m_shear_rate(shear_rate), m_offset(offset), m_dt(dt) {}
    double getShearRate(unsigned int) override { return m_shear_rate; }
    double getStrain(unsigned int t) override { return m_shear_rate * detail::since(t, m_offset) * m_dt; }
    unsigned int getOffset() override { return m_offset; }
private:
    const double m_shear_rate;
    const unsigned int m_offset;
    const double m_dt;
};

// exponential chirp: omega(t) = w0 (wf/w0)^{t/T}, phase = T w0 / ln(wf/w0) ((wf/w0)^{t/T} - 1)   (SpecificShearFunction.h:99-117)
class ChirpShearFunction : public ShearFunction {
public:
    ChirpShearFunction(double amp, double omega_0, double omega_f, double periodT, unsigned int offset, double dt)
        : m_amp(amp), m_omega_0(omega_0), m_omega_f(omega_f), m_periodT(periodT), m_offset(offset), m_dt(dt) {}
    double getShearRate(unsigned int t) override { return m_amp * omega(t) * std::cos(phase(t)); }
    double getStrain(unsigned int t) override { return m_amp * std::sin(phase(t)); }
    unsigned int getOffset() override { return m_offset; }
private:
    double omega(unsigned int t) const {
        return m_omega_0 * std::exp(m_dt * detail::since(t, m_offset) * std::log(m_omega_f / m_omega_0) / m_periodT);
    }
    double phase(unsigned int t) const {
        const double lg = std::log(m_omega_f / m_omega_0);
        return m_periodT * m_omega_0 / lg * (std::exp(m_dt * detail::since(t, m_offset) * lg / m_periodT) - 1);
    }
    const double m_amp, m_omega_0, m_omega_f, m_periodT;
    const unsigned int m_offset;
    const double m_dt;
};

// Tukey (tapered cosine) window: getStrain is the window value, getShearRate its time derivative
// (SpecificShearFunction.h:151-180)
class TukeyWindowFunction : public ShearFunction {
public:
    TukeyWindowFunction(double periodT, double tukey_param, unsigned int offset, double dt)
        : m_periodT(periodT), m_tukey_param(tukey_param), m_offset(offset), m_dt(dt), m_omega_value(2 * detail::kPi / tukey_param) {}
    double getShearRate(unsigned int t) override {
        const double r = detail::since(t, m_offset) * m_dt / m_periodT, a = m_tukey_param / 2;
        if (r <= 0 || r >= 1) return 0;
        if (r >= a && r <= 1 - a) return 0;
        if (r < 0.5) return -(std::sin(m_omega_value * (r - a))) / 2 * m_omega_value / m_periodT;
        return -(std::sin(m_omega_value * (r - 1 + a))) / 2 * m_omega_value / m_periodT;
    }
    double getStrain(unsigned int t) override {
        const double r = detail::since(t, m_offset) * m_dt / m_periodT, a = m_tukey_param / 2;
        if (r <= 0 || r >= 1) return 0;
        if (r >= a && r <= 1 - a) return 1;
        if (r < 0.5) return (1 + std::cos(m_omega_value * (r - a))) / 2;
        return (1 + std::cos(m_omega_value * (r - 1 + a))) / 2;
    }
    unsigned int getOffset() override { return m_offset; }
private:
    const double m_periodT, m_tukey_param;
    const unsigned int m_offset;
    const double m_dt, m_omega_value;
};

// strain = base * window, rate by the product rule (SpecificShearFunction.h:210-220)
class WindowedFunction : public ShearFunction {
public:
    WindowedFunction(std::shared_ptr<ShearFunction> base, std::shared_ptr<ShearFunction> window) : m_base(base), m_window(window) {}
    double getShearRate(unsigned int t) override {
        return m_base->getShearRate(t) * m_window->getStrain(t) + m_base->getStrain(t) * m_window->getShearRate(t);
    }
    double getStrain(unsigned int t) override { return m_base->getStrain(t) * m_window->getStrain(t); }
    unsigned int getOffset() override { return m_base->getOffset(); }
private:
    const std::shared_ptr<ShearFunction> m_base, m_window;
};

// HOOMD's Variant interface, reduced to what the path uses
class Variant {
public:
    virtual ~Variant() = default;
    virtual double getValue(unsigned int timestep) = 0;
};
class VariantConst : public Variant {
public:
    explicit VariantConst(double v) : m_v(v) {}
    double getValue(unsigned int) override { return m_v; }
private:
    double m_v;
};

// wrapped strain for the box tilt (VariantShearFunction.h:46-48, VariantShearFunction.cc:17-43)
class VariantShearFunction : public Variant {
public:
    VariantShearFunction(std::shared_ptr<ShearFunction> f, unsigned int total_timestep, double min_value, double max_value)
        : m_f(f), m_total(total_timestep), m_min(min_value), m_max(max_value), m_offset(f->getOffset()), m_range(max_value - min_value) {
        m_end = wrapValue(m_f->getStrain(m_offset + m_total));
    }
    double getValue(unsigned int t) override {
        if (t < m_offset) return 0;
        if (t >= m_offset + m_total) return m_end;
        return wrapValue(m_f->getStrain(t));
    }
    double wrapValue(double v) const { return v - m_range * std::floor((v - m_min) / m_range); }
private:
    const std::shared_ptr<ShearFunction> m_f;
    const unsigned int m_total;
    const double m_min, m_max;
    const unsigned int m_offset;
    double m_end, m_range;
};

}  // namespace pse_host
