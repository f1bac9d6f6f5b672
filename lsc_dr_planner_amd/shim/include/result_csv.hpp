// result_csv.hpp — the simulation result log in the reference's format, so that the reference's replay tool
// (MultiSyncReplayer::readCSVFile, src/multi_sync_replayer.cpp:54-110) reads what this build produces.
//
// Mirrors MultiSyncSimulator::saveSimulationResultAsCSV (reference src/multi_sync_simulator.cpp:586-656): one header
// line with "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time" per agent, then one line per sample time with those twelve
// fields per agent, numbers in std::ostream's default formatting (6 significant digits).  Missions with obstacles (mission.on != 0)
// carry "obs_id,t,px,py,pz,size" per obstacle behind the agents' columns (:603-610, :638-652): the obstacle MODELS are out of scope,
// the six fields are plain numbers the caller hands in (ObstacleSample).  SimulationSummaryCsv is the one-line-per-mission summary of
// saveSummarizedResultAsCSV (:658-709) with the reference's own column names.
#pragma once
#include <cstddef>
#include <fstream>
#include <functional>
#include <ostream>
#include <string>
#include <stdexcept>
#include <vector>

#include "sp_const.hpp"
#include "trajectory.hpp"

namespace DynamicPlanning {

// What the result log holds of an obstacle at one sample time (Obstacle::position, Obstacle::radius; :639-645)
struct ObstacleSample {
    double px = 0, py = 0, pz = 0, size = 0;
};

class SimulationResultCsv {
public:
    SimulationResultCsv(std::ostream& os, size_t qn, size_t on = 0) : os_(os), qn_(qn), on_(on) {}

    void writeHeader() {  // :590-610
        for (size_t qi = 0; qi < qn_; qi++)
            os_ << "id,t,px,py,pz,vx,vy,vz,ax,ay,az,planning_time" << ((qi + 1 < qn_ || on_ != 0) ? "," : "\n");
        for (size_t oi = 0; oi < on_; oi++) os_ << "obs_id,t,px,py,pz,size" << (oi + 1 < on_ ? "," : "\n");
    }

    // one sample time: the agents' states and their last total planning time (:616-636), then the obstacles (:638-652)
    // (a writer constructed with obstacle columns needs a sample per obstacle: std::invalid_argument otherwise -- the columns of a row
    // cannot be left out, and reading past a short vector is not an option)
    void writeRow(double t, const std::vector<State>& states, const std::vector<double>& planning_time,
                  const std::vector<ObstacleSample>& obstacles = {}) {
        if (states.size() < qn_ || planning_time.size() < qn_ || obstacles.size() < on_)
            throw std::invalid_argument("SimulationResultCsv::writeRow: fewer states / planning times / obstacle samples than the header's columns");
        for (size_t qi = 0; qi < qn_; qi++) {
            const State& s = states[qi];
            os_ << qi << "," << t << "," << s.position.x() << "," << s.position.y() << "," << s.position.z() << "," << s.velocity.x() << ","
                << s.velocity.y() << "," << s.velocity.z() << "," << s.acceleration.x() << "," << s.acceleration.y() << ","
                << s.acceleration.z() << "," << planning_time[qi] << ((qi + 1 < qn_ || on_ != 0) ? "," : "\n");
        }
        for (size_t oi = 0; oi < on_; oi++) {
            const ObstacleSample& o = obstacles[oi];
            os_ << oi << "," << t << "," << o.px << "," << o.py << "," << o.pz << "," << o.size << (oi + 1 < on_ ? "," : "\n");
        }
    }

    // one planned step starting at simulation time t: samples future_time = 0, record_time_step, ... < time_step of every
    // agent's current trajectory (AgentManager::getFutureState = desired_traj.getStateAt), :612-653
    // obstacle_at(oi, future_time): what ObstacleGenerator::getObstacle(oi) holds at that sample (the caller's obstacle model).  NOTE the
    // reference logs obstacle_generator.getObstacle(oi) UNCHANGED for every future_time of one step (:638-652 sit inside the sample loop
    // but the generator is only advanced once per step, :279): a callback that reproduces the reference's file returns the step's sample
    // whatever future_time is; one that advances the obstacle writes a finer -- different -- log.
    void writeStep(double t, double time_step, double record_time_step, const std::vector<traj_t>& trajs,
                   const std::vector<double>& planning_time, const std::function<ObstacleSample(size_t, double)>& obstacle_at = nullptr) {
        double future_time = 0;
        std::vector<State> states(qn_);
        std::vector<ObstacleSample> obs(on_);
        while (future_time < time_step) {
            for (size_t qi = 0; qi < qn_; qi++) states[qi] = trajs[qi].getStateAt(future_time);
            for (size_t oi = 0; oi < on_; oi++) obs[oi] = obstacle_at ? obstacle_at(oi, future_time) : ObstacleSample();
            writeRow(t, states, planning_time, obs);
            future_time += record_time_step;
            t += record_time_step;
        }
    }

private:
    std::ostream& os_;
    size_t qn_, on_;
};

// One line per mission: MultiSyncSimulator::saveSummarizedResultAsCSV (:658-709), the reference's column names and order.  The time
// statistics are those of PlanningTimeStatistics (include/sp_const.hpp); `traj_optimization_time` is where a GPU build's QP time goes.
struct SimulationSummary {
    std::string start_time;  // mission_start_time (a string in the reference: ROS time with six decimals)
    double total_flight_time = 0, total_flight_distance = 0, safety_ratio_agent = 0, safety_ratio_obs = 0, vel_excess_ratio = 0, acc_excess_ratio = 0;
    double mapf_time_average = 0, mapf_time_min = 0, mapf_time_max = 0;
    double planning_time_average = 0, planning_time_min = 0, planning_time_max = 0;
    double initial_traj_planning_time = 0, obstacle_prediction_time = 0, goal_planning_time = 0, lsc_generation_time = 0, sfc_generation_time = 0,
           traj_optimization_time = 0;
    std::string mission_file_name, world_file_name, planner_mode, goal_mode, mapf_mode;
    double communication_range = 0;
    int world_dimension = 0, M = 0;
    double dt = 0;
};

class SimulationSummaryCsv {
public:
    static void writeDescription(std::ostream& os) {  // :669-679
        os << "start_time,total_flight_time,total_flight_distance,"
           << "safety_ratio_agent,safety_ratio_obs,"
           << "vel_excess_ratio,acc_excess_ratio,"
           << "mapf_time_average,mapf_time_min,mapf_time_max,"
           << "planning_time_average,planning_time_min,planning_time_max,"
           << "initial_traj_planning_time,obstacle_prediction_time,goal_planning_time,"
           << "lsc_generation_time,sfc_generation_time,traj_optimization_time,"
           << "mission_file_name,world_file_name,"
           << "planner_mode,goal_mode,mapf_mode,"
           << "communication_range,world_dimension,M,dt\n";
    }
    static void writeRow(std::ostream& os, const SimulationSummary& s) {  // :681-708
        os << s.start_time << "," << s.total_flight_time << "," << s.total_flight_distance << "," << s.safety_ratio_agent << "," << s.safety_ratio_obs << ","
           << s.vel_excess_ratio << "," << s.acc_excess_ratio << "," << s.mapf_time_average << "," << s.mapf_time_min << "," << s.mapf_time_max << ","
           << s.planning_time_average << "," << s.planning_time_min << "," << s.planning_time_max << "," << s.initial_traj_planning_time << ","
           << s.obstacle_prediction_time << "," << s.goal_planning_time << "," << s.lsc_generation_time << "," << s.sfc_generation_time << ","
           << s.traj_optimization_time << "," << s.mission_file_name << "," << s.world_file_name << "," << s.planner_mode << "," << s.goal_mode << ","
           << s.mapf_mode << "," << s.communication_range << "," << s.world_dimension << "," << s.M << "," << s.dt << "\n";
    }
    // the reference appends to log/summary_<name>.csv and writes the description only into a new or empty file (:659-668)
    static bool append(const std::string& file_name, const SimulationSummary& s) {
        bool print_description = false;
        {
            std::ifstream in(file_name);
            if (!in || in.peek() == std::ifstream::traits_type::eof()) print_description = true;
        }
        std::ofstream out(file_name, std::ios_base::app);
        if (!out) return false;
        if (print_description) writeDescription(out);
        writeRow(out, s);
        return true;
    }
};

}  // namespace DynamicPlanning
