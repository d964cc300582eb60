// od-msspe-hip: command-line front end with od-msspe's flags, environment variables, CSV output
// and coverage report (/root/reference/od-msspe/src/main.rs:596-861, config.rs:11-148), driving
// libmsspe_hip.so for the hot path.
#include <cstdio>
#include <exception>

#include "od_msspe.hpp"

int main(int argc, char **argv)
{
    try {
        const od_msspe::Args args = od_msspe::Args::parse(argc, argv);
        std::string report;
        const int rc = od_msspe::run(args, report);
        std::fputs(report.c_str(), stdout);
        return rc;
    } catch (const od_msspe::UsageError &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 2;   // clap's usage-error exit status
    } catch (const od_msspe::Panic &e) {
        std::fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
        return 101;   // Rust's panic exit status
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
