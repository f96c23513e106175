"""The spine's finite-state machine (upkie/cpp/spine/StateMachine.{h,cpp}):
stop -> (>= 5 stop cycles, then kStart) -> reset -> idle <-> step, kStop from
idle goes back to stop; invalid requests answer `Request.kError`."""

from enum import IntEnum

from .request import Request

kNbStopCycles = 5  # StateMachine.h:10


class State(IntEnum):
    kSendStops = 0
    kReset = 1
    kIdle = 2
    kStep = 3
    kShutdown = 4
    kOver = 5


class Event(IntEnum):
    kCycleBeginning = 0
    kCycleEnd = 1
    kInterrupt = 2


class StateMachine:
    def __init__(self, interface):
        self._interface = interface
        self._state = State.kSendStops
        self._stop_cycles = 0
        self._enter_state(State.kSendStops)  # sets the request to kNone, StateMachine.cpp:14-17

    @property
    def state(self) -> State:
        return self._state

    def is_over_after_this_cycle(self) -> bool:  # StateMachine.h:97-99
        return self._state == State.kShutdown and self._stop_cycles + 1 == kNbStopCycles

    def process_event(self, event: Event) -> None:  # StateMachine.cpp:19-37
        if event == Event.kInterrupt:
            if self._state != State.kShutdown:
                self._enter_state(State.kShutdown)
        elif event == Event.kCycleBeginning:
            self._process_cycle_beginning()
        elif event == Event.kCycleEnd:
            self._process_cycle_end()
        else:
            self._enter_state(State.kShutdown)

    def _process_cycle_beginning(self) -> None:  # StateMachine.cpp:39-103
        request = self._interface.request()
        if self._state == State.kIdle:
            if request == Request.kNone:
                pass
            elif request == Request.kAction:
                self._enter_state(State.kStep)
            elif request == Request.kStart:  # invalid from idle: stop the spine first
                self._enter_state(State.kIdle)  # resets the request
            elif request == Request.kStop:
                self._enter_state(State.kSendStops)
            elif request != Request.kError:
                self._interface.set_request(Request.kError)
        elif self._state == State.kSendStops:
            if request == Request.kNone:
                pass
            elif request == Request.kAction:
                self._interface.set_request(Request.kError)
            elif request == Request.kStart:
                if self._stop_cycles >= kNbStopCycles:
                    self._enter_state(State.kReset)
            elif request == Request.kStop:
                self._enter_state(State.kSendStops)
            elif request != Request.kError:
                # (the C++ default branch rewrites kError over kError; not
                # rewriting it is the same protocol without the window in which
                # the agent's acknowledgement kError -> kNone gets overwritten)
                self._interface.set_request(Request.kError)
        # kReset / kStep: a cycle beginning should not happen; kShutdown / kOver: nothing

    def _process_cycle_end(self) -> None:  # StateMachine.cpp:105-135
        if self._state == State.kReset:
            self._enter_state(State.kIdle)
        elif self._state == State.kSendStops:
            self._stop_cycles += 1
        elif self._state == State.kStep:
            self._enter_state(State.kIdle)
        elif self._state == State.kShutdown:
            self._stop_cycles += 1
            if self._stop_cycles == kNbStopCycles:
                self._enter_state(State.kOver)

    def _enter_state(self, next_state: State) -> None:  # StateMachine.cpp:137-160
        if next_state == State.kIdle:
            self._interface.set_request(Request.kNone)
        elif next_state in (State.kSendStops, State.kShutdown):
            self._interface.set_request(Request.kNone)
            self._stop_cycles = 0
        self._state = next_state
