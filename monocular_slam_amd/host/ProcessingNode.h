// Stage base class and frame-by-frame pipeline of the reference (src/ProcessingNode.h:16-32, src/Pipeline.h:22-67),
// re-stated: same virtuals, same call order (init() on frame 0, then validationCheck -> process per node; a failed
// check skips the remaining stages of THIS frame), same ownership (the pipeline deletes its nodes).
#pragma once
#include "DataManager.h"

class ProcessingNode {
public:
    explicit ProcessingNode(std::string name) : name(std::move(name)) {}
    virtual ~ProcessingNode() {}
    virtual void init() {}
    virtual void finish() {}
    virtual void destroy() {}
    // batch form (unused by the fused pipeline)
    virtual void process(DataManager&) {}
    virtual bool validationCheck(DataManager&) { return true; }
    // fused form
    virtual void process(DataManager&, int /*frameIdx*/) {}
    virtual bool validationCheck(DataManager&, int /*frameIdx*/) { return true; }
    std::string name;
};

class BaseProcessingPipeline {
public:
    std::vector<ProcessingNode*> stages;
    void addStage(ProcessingNode* engine) { stages.push_back(engine); }
    virtual ~BaseProcessingPipeline() {
        for (ProcessingNode* n : stages) { n->destroy(); delete n; }
    }
};

class ProcessingPipeline : public BaseProcessingPipeline {
public:
    void process(DataManager& data, int frameIdx) {
        if (frameIdx == 0)
            for (ProcessingNode* n : stages) n->init();
        for (ProcessingNode* n : stages) {
            if (!n->validationCheck(data, frameIdx)) {
                std::cout << "Node [" << n->name << "]: validation check failed." << std::endl;
                return;
            }
            n->process(data, frameIdx);
        }
    }
};
