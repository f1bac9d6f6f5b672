// result_csv.hpp — the simulation result log in the reference's format, so that the reference's replay tool
// (MultiSyncReplayer::readCSVFile, src/multi_sync_replayer.cpp:54-110) reads what this build produces.
//
// Mirrors MultiSyncSimulator::saveSimulationResultAsCSV (reference src/multi_sync_simulator.cpp:586-656): one header
// line with "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time" per agent, then one line per sample time with those twelve
// fields per agent, numbers in std::ostream's default formatting (6 significant digits).  Obstacle columns (:603-610,
// :638-652) belong to the obstacle models, which are out of scope: missions without obstacles only (mission.on == 0).
#pragma once
#include <cstddef>
#include <ostream>
#include <vector>

#include "sp_const.hpp"
#include "trajectory.hpp"

namespace DynamicPlanning {

class SimulationResultCsv {
public:
    SimulationResultCsv(std::ostream& os, size_t qn) : os_(os), qn_(qn) {}

    void writeHeader() {  // :590-601
        for (size_t qi = 0; qi < qn_; qi++) os_ << "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time" << (qi + 1 < qn_ ? "," : "\n");
    }

    // one sample time: the agents' states and their last total planning time (:616-636)
    void writeRow(double t, const std::vector<State>& states, const std::vector<double>& planning_time) {
        for (size_t qi = 0; qi < qn_; qi++) {
            const State& s = states[qi];
            os_ << qi << "," << t << "," << s.position.x() << "," << s.position.y() << "," << s.position.z() << "," << s.velocity.x() << ","
                << s.velocity.y() << "," << s.velocity.z() << "," << s.acceleration.x() << "," << s.acceleration.y() << ","
                << s.acceleration.z() << "," << planning_time[qi] << (qi + 1 < qn_ ? "," : "\n");
        }
    }

    // one planned step starting at simulation time t: samples future_time = 0, record_time_step, ... < time_step of every
    // agent's current trajectory (AgentManager::getFutureState = desired_traj.getStateAt), :612-653
    void writeStep(double t, double time_step, double record_time_step, const std::vector<traj_t>& trajs,
                   const std::vector<double>& planning_time) {
        double future_time = 0;
        std::vector<State> states(qn_);
        while (future_time < time_step) {
            for (size_t qi = 0; qi < qn_; qi++) states[qi] = trajs[qi].getStateAt(future_time);
            writeRow(t, states, planning_time);
            future_time += record_time_step;
            t += record_time_step;
        }
    }

private:
    std::ostream& os_;
    size_t qn_;
};

}  // namespace DynamicPlanning
