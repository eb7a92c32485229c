#!/bin/bash
# scratch: the command of the last gpurun call (tools/*.sh hold the reusable recipes)
